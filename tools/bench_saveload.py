"""save / load timing of a 10^8-k-mer model (header + km.bin + rest.bin on /tmp)."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmcex_amd import KModel, synth_torch
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
km, cnt = synth_torch.make_stream(n, 31, 1, 1023, dev)
m = KModel(1, 1023, 7, 5)
m.build_dev(31, km.data_ptr(), cnt.data_ptr(), km.numel())
d = tempfile.mkdtemp(dir="/tmp")
for rep in range(2):
    t = time.time(); m.save(d); ts = time.time() - t
    sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    t = time.time(); m2 = KModel.load(d); tl = time.time() - t
    print(f"save {ts*1e3:.0f} ms ({sz/ts/1e9:.2f} GB/s, {sz/1e6:.0f} MB)  load {tl*1e3:.0f} ms ({sz/tl/1e9:.2f} GB/s)", flush=True)
    del m2
shutil.rmtree(d)
