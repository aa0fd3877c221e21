"""Soak ONE configuration: build it many times under a hook setting and compare every build with the CPU oracle (arrays by
digest).  Finds timing-dependent results.  usage: python tools/soak_case.py k ci cs nh nb n seed forced iters [ENV=VAL ...]"""
import os as _os; _os.environ.setdefault("KMX_TEST_HOOKS", "1")   # forced code paths are test hooks
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from kmcex_amd import KModel, synth
k, ci, cs, nh, nb, n, seed, forced, iters = (int(x) for x in sys.argv[1:10])
for kv in sys.argv[10:]:
    a, b = kv.split("="); os.environ[a] = b
km, cnt = synth.make_stream(n, k, ci, cs, seed_k=seed, seed_c=seed + 1)
if forced: cnt = np.maximum(cnt, ci + 3).astype(np.uint32)
o = O.OracleModel(ci, cs, nh, nb); o.build(k, km, cnt)
def dig(x): return hashlib.sha1(np.ascontiguousarray(x).tobytes()).hexdigest()
want = [dig(o.array_bytes(w, a)) for a in range(nb) for w in ("tag", "value")] + [dig(o.array_bytes("km_back"))]
bad = 0; t0 = time.time()
for it in range(iters):
    m = KModel(ci, cs, nh, nb); m.build_packed(k, km, cnt)
    got = [dig(m.download(w, a)) for a in range(nb) for w in ("tag", "value")] + [dig(m.download("km_back"))]
    if got != want:
        bad += 1
        print(f"iteration {it}: differs in {[i for i in range(len(want)) if got[i] != want[i]]} (index = 2*array + (0 tag, 1 value), last = km_back)", flush=True)
        st, so = m.stats(), o.stats()
        print("   stats dev/oracle: attempts", st.attempts, so.attempts, "successes", st.successes, so.successes, "rest", st.rest_entries, so.rest_entries, "contended", st.contended, "fin_iters", st.finisher_iters, flush=True)
        for a in range(nb):
            for w in ("tag", "value"):
                d, r = np.unpackbits(m.download(w, a)), np.unpackbits(o.array_bytes(w, a))
                if (d != r).any(): print(f"   {w} {a}: bits only on the device {int((d & ~r).sum())}, only in the oracle {int((r & ~d).sum())}, positions (device only) {np.flatnonzero(d & ~r)[:12].tolist()} (oracle only) {np.flatnonzero(r & ~d)[:12].tolist()}", flush=True)
    del m
    if it % 500 == 499: print(f"[{time.time()-t0:.0f}s] {it+1} builds, {bad} wrong", flush=True)
print(f"SOAK {sys.argv[1:]}: {iters} builds, {bad} wrong")
