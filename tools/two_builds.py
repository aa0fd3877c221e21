"""How much do two HIP streams overlap on this chip?  Two independent models, each on its own stream and host thread,
built side by side (the ordered insert leaves the memory system idle in its latency-bound steps: a second build can
fill them) against the same two builds one after the other.  usage: python tools/two_builds.py [k-mers per model]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kmcex_amd import KModel, synth_torch

import ctypes
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
# optional: CUs of stream 0 (the rest go to stream 1) -> the two builds run on disjoint sets of compute units
# (hipExtStreamCreateWithCUMask; mask bit i = CU i / 8 of XCD i % 8 on this chip)
cu_split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(lo, hi, total=256):
    words = (ctypes.c_uint32 * (total // 32))()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = _hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(total // 32), words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed: {rc}"
    return st.value
dev = torch.device("cuda", 0)
k, ci, cs, nh, nb = 31, 1, 1023, 7, 5
data, models, streams = [], [], []
for r in range(2):
    km, cnt = synth_torch.make_stream(n, k, ci, cs, dev, seed_k=1 + 100 * r, seed_c=2 + 100 * r)
    m = KModel(ci, cs, nh, nb)
    if cu_split:
        torch.cuda.init(); torch.zeros(1, device=dev)
        st = masked_stream(0, cu_split) if r == 0 else masked_stream(cu_split, 256)
        m.set_stream(st)
    else:
        st = torch.cuda.Stream(device=dev)
        m.set_stream(st.cuda_stream)
    data.append((km, cnt)); models.append(m); streams.append(st)

def build(r):
    km, cnt = data[r]
    models[r].build_dev(k, km.data_ptr(), cnt.data_ptr(), km.numel())

for r in range(2):
    build(r)                                                   # warm-up (allocations)
torch.cuda.synchronize()
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    for r in range(2):
        build(r)
torch.cuda.synchronize()
t_seq = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps):
    th = [threading.Thread(target=build, args=(r,)) for r in range(2)]
    for t in th: t.start()
    for t in th: t.join()
torch.cuda.synchronize()
t_par = (time.perf_counter() - t0) / reps
tot = sum(d[0].numel() for d in data)
if cu_split:
    for r in range(2):                                         # each build alone on its share of the chip
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): build(r)
        torch.cuda.synchronize()
        print(f"  build {r} alone on {cu_split if r == 0 else 256 - cu_split} CUs: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms")
print(f"CU split {cu_split or 'none'}: 2 x {n} k-mers: one after the other {t_seq * 1e3:.1f} ms ({tot / t_seq / 1e9:.3f} G k-mers/s), side by side on two streams {t_par * 1e3:.1f} ms ({tot / t_par / 1e9:.3f} G k-mers/s)")
