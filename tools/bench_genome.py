"""Throughput on a genome-like stream (overlapping k-mers of a random sequence): the neighbour path of the query is live."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kmcex_amd import KModel, synth
n_bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30_000_000
k, ci, cs, nh, nb = 31, 1, 1023, 7, 5
t = time.time(); km, cnt = synth.genome_stream(n_bases, k, ci, cs); print(f"{len(cnt)} k-mers ({time.time()-t:.1f}s host)", flush=True)
dev = torch.device("cuda", 0)
dk = torch.from_numpy(km.view(np.int64)).to(dev); dc = torch.from_numpy(cnt.view(np.int32)).to(dev)
m = KModel(ci, cs, nh, nb); m.set_stream(torch.cuda.current_stream().cuda_stream)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time(); m.build_dev(k, dk.data_ptr(), dc.data_ptr(), len(cnt)); torch.cuda.synchronize(); dt = time.time() - t
    print(f"insert: {len(cnt)/dt/1e6:.1f} M k-mers/s", flush=True)
out = torch.empty(len(cnt), dtype=torch.int32, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time(); m.kmer_to_occ_dev(dk.data_ptr(), len(cnt), out.data_ptr()); torch.cuda.synchronize(); dt = time.time() - t
    print(f"query (stored k-mers): {len(cnt)/dt/1e6:.1f} M k-mers/s", flush=True)
# successors of stored k-mers: 3 of 4 absent, 1 of 4 present
s = dk[: len(cnt)//2]; q = ((s << 2) | 1) & ((1 << 62) - 1)
out2 = torch.empty(q.numel(), dtype=torch.int32, device=dev)
torch.cuda.synchronize(); t = time.time(); m.kmer_to_occ_dev(q.data_ptr(), q.numel(), out2.data_ptr()); torch.cuda.synchronize(); dt = time.time() - t
print(f"query (successor k-mers): {q.numel()/dt/1e6:.1f} M k-mers/s, nonzero {(out2!=0).float().mean().item():.3f}")
