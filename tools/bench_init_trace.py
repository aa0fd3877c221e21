"""KModel::init(db) end to end with its phase times (KMX_INIT_TRACE=1 prints them), on the layouts and record widths real databases
have: usage: python tools/bench_init_trace.py [n_kmers] [reps] [layout=kmc1|kmc2] [k=31|55] [bins=512]
k = 31: the bench's stream, written from the device; k = 55 (two-word k-mers, nh 9 nb 6 cs 4095, 15-byte records): a host stream."""
import os, shutil, sys, tempfile, time
os.environ.setdefault("KMX_INIT_TRACE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from kmcex_amd import KModel, kmcdb, synth, synth_torch
opt = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
pos = [a for a in sys.argv[1:] if "=" not in a]
n = int(float(pos[0])) if pos else 100_000_000
reps = int(pos[1]) if len(pos) > 1 else 4
layout, k, bins = opt.get("layout", "kmc1"), int(opt.get("k", 31)), int(opt.get("bins", 512))
ci, cs, nh, nb = (1, 1023, 7, 5) if k <= 31 else (1, 4095, 9, 6)
dev = torch.device("cuda", 0)
base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
tmp = tempfile.mkdtemp(prefix="kmx_init_", dir=base)
try:
    db = os.path.join(tmp, "db")
    if k <= 31:
        km, cnt = synth_torch.make_stream(n, k, ci, cs, dev)
        (bench.write_kmc1_from_device if layout == "kmc1" else lambda *a: bench.write_kmc2_from_device(*a, n_bins=bins))(db, km, cnt, k, ci, cs)
        n_db = km.numel()
        del km, cnt
    else:
        t = time.time(); km, cnt = synth.make_stream(n, k, ci, cs); print(f"host stream of {len(cnt)} {k}-mers in {time.time() - t:.1f} s", flush=True)
        (kmcdb.write_kmc1 if layout == "kmc1" else lambda *a: kmcdb.write_kmc2(*a, n_bins=bins))(db, km, cnt, k, ci, cs)
        n_db = len(cnt)
        del km, cnt
    print(f"{layout} database, k = {k}: {n_db} records, {os.path.getsize(db + '.kmc_suf')} + {os.path.getsize(db + '.kmc_pre')} bytes", flush=True)
    m = KModel(ci, cs, nh, nb)
    for rep in range(reps):
        t = time.perf_counter(); m.init(db); dt = time.perf_counter() - t
        print(f"init(db) {layout} k={k} rep {rep}: {n_db / dt / 1e6:.1f} M k-mers/s end to end ({dt * 1e3:.1f} ms)", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
