"""One-off randomized stress of the ordered insert against the CPU oracle: many small and mid-size configurations,
biased to heavy contention (everything into the coupled arrays, tiny arrays), every forced code path.
usage: python tools/stress_parity.py [seconds] [seed] [big|small]   (small: at most 30 000 k-mers, for libraries built to overflow)"""
import os as _os; _os.environ.setdefault("KMX_TEST_HOOKS", "1")   # forced code paths are test hooks
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from kmcex_amd import KModel, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
HOOKS = [dict(), dict(KMX_NSUB0="0", KMX_NSUB1="0"), dict(KMX_NSUB0="1", KMX_NSUB1="1"), dict(KMX_NSUB0="3", KMX_NSUB1="2"),
         dict(KMX_FIN_GLOBAL="1"), dict(KMX_NSUB0="2", KMX_RESOLVE_GATHER="1"), dict(KMX_KMB_DIRECT="1"), dict(KMX_KMB_DIRECT="1", KMX_NSUB0="1"),
         dict(KMX_PIPE="0"), dict(KMX_PIPE="8"), dict(KMX_PIPE="3", KMX_NSUB0="1", KMX_NSUB1="1"), dict(KMX_PIPE="0", KMX_FIN_GLOBAL="1"), dict(KMX_KMB_HOST="0"), dict(KMX_KMB_HOST="0", KMX_PIPE="0"),
         dict(KMX_SMALL_DETECT="1"), dict(KMX_SMALL_DETECT="0"), dict(KMX_SMALL_DETECT="1", KMX_NSUB0="1", KMX_NSUB1="1"), dict(KMX_SMALL_DETECT="1", KMX_PIPE="0")]   # the big rounds unpipelined / list by list / in 3 groups; km_back emission in launches of its own
SIZES = [600000, 1500000, 3000000, 5000000] if "big" in sys.argv else ([40, 300, 3000, 12000, 30000] if "small" in sys.argv else [40, 300, 3000, 30000, 120000, 300000, 700000])   # big: several blocks, final partial block
t0 = time.time(); done = 0; contended = 0
while time.time() - t0 < budget:
    k = int(rng.integers(12, 65)); nh = int(rng.integers(3, 17)); nb = int(rng.integers(1, 9))
    ci = int(rng.choice([1, 1, 2, 3])); cs = int(max(1 << nh, ci + 3) + rng.integers(0, 2000))
    n = int(rng.choice(SIZES))
    if 4 ** min(k, 20) < 8 * n: continue
    try:
        o = O.OracleModel(ci, cs, nh, nb)
    except ValueError:
        continue
    seed = int(rng.integers(1, 1 << 30))
    km, cnt = synth.make_stream(n, k, ci, cs, seed_k=seed, seed_c=seed + 1)
    if rng.random() < 0.6: cnt = np.maximum(cnt, ci + 3).astype(np.uint32)     # everything into the coupled arrays
    o.build(k, km, cnt); so = o.stats()
    g1 = int(rng.integers(1, nh + 1)); g2 = int(rng.integers(g1, nh + 1))       # groups the check fetches its positions in (forced path only)
    for env in [dict(HOOKS[int(rng.integers(0, len(HOOKS)))], KMX_NH_FIRST=str(g1), KMX_NH_SECOND=str(g2)), HOOKS[0]]:
        for v in ("KMX_NSUB0", "KMX_NSUB1", "KMX_FIN_GLOBAL", "KMX_RESOLVE_GATHER", "KMX_KMB_DIRECT", "KMX_PIPE", "KMX_KMB_HOST", "KMX_NH_FIRST", "KMX_NH_SECOND", "KMX_SMALL_DETECT"): os.environ.pop(v, None)
        os.environ.update(env)
        m = KModel(ci, cs, nh, nb); m.build_packed(k, km, cnt); st = m.stats()
        tag = (k, ci, cs, nh, nb, n, seed, env)
        for a in range(nb):
            assert np.array_equal(m.download("tag", a), o.array_bytes("tag", a)), ("tag", a, tag)
            assert np.array_equal(m.download("value", a), o.array_bytes("value", a)), ("value", a, tag)
            assert not m.download("claims", a).any(), ("claims", a, tag)
        assert np.array_equal(m.download("km_back"), o.array_bytes("km_back")), ("km_back", tag)
        for f in range(st.bf_num):
            assert np.array_equal(m.download("bf", f), o.array_bytes("bf", f)) and np.array_equal(m.download("bf_back", f), o.array_bytes("bf_back", f)), ("bloom", f, tag)
        assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries), ("stats", tag)
        contended += st.contended
        del m
    done += 1
    if done % (5 if "big" in sys.argv else 20) == 0: print(f"[{time.time()-t0:.0f}s] {done} configurations, {contended} contended k-mers decided, all bit-exact", flush=True)
print(f"STRESS OK: {done} configurations x 2 code paths, {contended} contended k-mers, {time.time()-t0:.0f}s")
