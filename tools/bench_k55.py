"""One-off data point for BASELINE configs[4]'s shape on ONE GPU: k=55 nh=9 nb=6 cs=4095 (two-word k-mers, NHM=16 kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kmcex_amd import KModel, synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
k, ci, cs, nh, nb = 55, 1, 4095, 9, 6
t = time.time(); km, cnt = synth.make_stream(n, k, ci, cs); print(f"host stream {len(cnt)} in {time.time()-t:.1f}s", flush=True)
dev = torch.device("cuda", 0)
dk = torch.from_numpy(km.view(np.int64)).to(dev); dc = torch.from_numpy(cnt.view(np.int32)).to(dev)
m = KModel(ci, cs, nh, nb)
m.set_stream(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    torch.cuda.synchronize(); t = time.time(); m.build_dev(k, dk.data_ptr(), dc.data_ptr(), len(cnt)); torch.cuda.synchronize(); dt = time.time() - t
    print(f"insert rep {rep}: {len(cnt)/dt/1e6:.1f} M k-mers/s ({dt*1e3:.1f} ms)", flush=True)
st = m.stats()
print("stats", st.attempts, st.successes, st.rest_entries, st.contended, st.finisher_iters)
out = torch.empty(len(cnt), dtype=torch.int32, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t = time.time(); m.kmer_to_occ_dev(dk.data_ptr(), len(cnt), out.data_ptr()); torch.cuda.synchronize(); dt = time.time() - t
    print(f"query rep {rep}: {len(cnt)/dt/1e6:.1f} M k-mers/s", flush=True)
m.set_profile(True); m.kernel_times(True); m.build_dev(k, dk.data_ptr(), dc.data_ptr(), len(cnt))
for kname, v in m.kernel_times(True).items():
    if v["launches"]: print(f"  {kname:14s} {v['seconds']*1e3:8.2f} ms {v['launches']} launches")
