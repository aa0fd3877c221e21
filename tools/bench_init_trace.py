"""KModel::init(db) phase times only (KMX_INIT_TRACE=1 prints them): usage: python tools/bench_init_trace.py [n_kmers] [reps]"""
import os, shutil, sys, tempfile, time
os.environ.setdefault("KMX_INIT_TRACE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kmcex_amd import KModel, synth_torch
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
km, cnt = synth_torch.make_stream(n, 31, 1, 1023, dev)
base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
tmp = tempfile.mkdtemp(prefix="kmx_init_", dir=base)
try:
    db = os.path.join(tmp, "db")
    bench.write_kmc1_from_device(db, km, cnt, 31, 1, 1023)
    m = KModel(1, 1023, 7, 5)
    for rep in range(reps):
        t = time.perf_counter(); m.init(db); dt = time.perf_counter() - t
        print(f"init(db) rep {rep}: {km.numel() / dt / 1e6:.1f} M k-mers/s end to end ({dt * 1e3:.1f} ms)", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
