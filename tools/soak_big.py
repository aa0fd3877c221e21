"""The bench configuration built over and over on the production path (default hooks): every build must leave the same
arrays (digests of tag/value arrays and km_back).  usage: python tools/soak_big.py [n] [builds]"""
import os as _os; _os.environ.setdefault("KMX_TEST_HOOKS", "1")   # forced code paths are test hooks
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kmcex_amd import KModel, synth_torch
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K, CI, CS, NH, NB = 31, 1, 1023, 7, 5
dev = torch.device("cuda", 0)
km, cnt = synth_torch.make_stream(n, K, CI, CS, dev)
m = KModel(CI, CS, NH, NB)
def digests():
    d = [hashlib.sha1(np.ascontiguousarray(m.download(w, a)).tobytes()).hexdigest() for a in range(NB) for w in ("tag", "value")]
    return d + [hashlib.sha1(np.ascontiguousarray(m.download("km_back")).tobytes()).hexdigest(), hashlib.sha1(np.ascontiguousarray(m.download("bf", 0)).tobytes()).hexdigest()]
first = None; bad = 0; t0 = time.time()
for it in range(builds):
    m.build_dev(K, km.data_ptr(), cnt.data_ptr(), km.numel())
    d = digests(); st = m.stats()
    if first is None: first = (d, st.attempts, st.successes, st.rest_entries)
    elif (d, st.attempts, st.successes, st.rest_entries) != first:
        bad += 1; print(f"build {it}: differs from build 0 in {[i for i in range(len(d)) if d[i] != first[0][i]]}", flush=True)
    if it % 5 == 4: print(f"[{time.time()-t0:.0f}s] {it+1} builds, {bad} different", flush=True)
print(f"SOAK {km.numel()} k-mers x {builds} builds: {bad} different from the first; successes {first[2]}, rest {first[3]}")
