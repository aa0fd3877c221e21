"""End-to-end KModel::init(db) from a KMC database in tmpfs (raw-record feed -> pinned H2D -> GPU decode -> insert).
usage: bench_init.py [n_kmers]   (KMX_CTRL_DEBUG=1 prints the phase times of every call)"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kmcex_amd import KModel, synth_torch
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
km, cnt = synth_torch.make_stream(n, 31, 1, 1023, dev)
base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
tmp = tempfile.mkdtemp(prefix="kmx_init_", dir=base)
try:
    db = os.path.join(tmp, "db")
    bench.write_kmc1_from_device(db, km, cnt, 31, 1, 1023)
    m = KModel(1, 1023, 7, 5)
    for rep in range(4):
        t = time.perf_counter(); m.init(db); dt = time.perf_counter() - t
        print(f"init(db) rep {rep}: {km.numel() / dt / 1e6:.1f} M k-mers/s end to end ({dt * 1e3:.1f} ms)", flush=True)
    m2 = KModel(1, 1023, 7, 5)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); m2.build_dev(31, km.data_ptr(), cnt.data_ptr(), km.numel()); torch.cuda.synchronize(); dt_dev = time.perf_counter() - t
    print(f"kmx_build_dev (listing resident in HBM): {km.numel() / dt_dev / 1e6:.1f} M k-mers/s ({dt_dev * 1e3:.1f} ms)", flush=True)
    assert all((m.download("tag", a) == m2.download("tag", a)).all() for a in range(5))
    print("same arrays as the resident build")
    # the host-pointer ABI (pageable caller buffers -> pinned double-buffered hipMemcpyAsync ring -> insert)
    import numpy as np
    hk, hc = km.cpu().numpy().view(np.uint64), cnt.cpu().numpy().view(np.uint32)
    m3 = KModel(1, 1023, 7, 5)
    for rep in range(3):
        t = time.perf_counter(); m3.build_packed(31, hk, hc); dt_host = time.perf_counter() - t
        print(f"kmx_build_host rep {rep}: {len(hc) / dt_host / 1e6:.1f} M k-mers/s ({dt_host * 1e3:.1f} ms) = {dt_host / dt_dev:.2f} x the resident build", flush=True)
    n_bf = [int((hc == 1).sum())]
    for rep in range(2):
        t = time.perf_counter(); m3.begin(31, n_bf, len(hc)); m3.insert_batch(hk, hc); m3.finish(); dt_str = time.perf_counter() - t
        print(f"kmx_begin + kmx_insert_batch (host pointers) + kmx_finish rep {rep}: {len(hc) / dt_str / 1e6:.1f} M k-mers/s ({dt_str * 1e3:.1f} ms) = {dt_str / dt_dev:.2f} x the resident build", flush=True)
    assert all((m3.download("tag", a) == m2.download("tag", a)).all() for a in range(5))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
