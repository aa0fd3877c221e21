"""Randomised configurations like tools/stress_parity.py, but every configuration is built many times under each hook
setting and every build is compared with the CPU oracle by digest: hunts timing-dependent results.
usage: python tools/soak_stress.py [seconds] [seed] [builds per hook setting]"""
import os as _os; _os.environ.setdefault("KMX_TEST_HOOKS", "1")   # forced code paths are test hooks
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from kmcex_amd import KModel, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
HOOKS = [dict(), dict(KMX_NSUB0="0", KMX_NSUB1="0"), dict(KMX_NSUB0="0", KMX_NSUB1="0", KMX_FIN_GLOBAL="1"), dict(KMX_NSUB0="1", KMX_NSUB1="1"),
         dict(KMX_NSUB0="2", KMX_NSUB1="2", KMX_FIN_GLOBAL="1"), dict(KMX_NSUB0="2", KMX_RESOLVE_GATHER="1"), dict(KMX_PIPE="0", KMX_NSUB1="0")]
VARS = ("KMX_NSUB0", "KMX_NSUB1", "KMX_FIN_GLOBAL", "KMX_RESOLVE_GATHER", "KMX_KMB_DIRECT", "KMX_PIPE", "KMX_KMB_HOST")
def dig(x): return hashlib.sha1(np.ascontiguousarray(x).tobytes()).hexdigest()
t0 = time.time(); done = 0; builds = 0; bad = 0
while time.time() - t0 < budget:
    k = int(rng.integers(12, 65)); nh = int(rng.integers(3, 9)); nb = int(rng.integers(1, 5))
    ci = int(rng.choice([1, 2, 3])); cs = int(max(1 << nh, ci + 3) + rng.integers(0, 300))
    n = int(rng.choice([3000, 20000, 60000, 120000, 250000]))
    if 4 ** min(k, 20) < 8 * n: continue
    try: o = O.OracleModel(ci, cs, nh, nb)
    except ValueError: continue
    seed = int(rng.integers(1, 1 << 30))
    km, cnt = synth.make_stream(n, k, ci, cs, seed_k=seed, seed_c=seed + 1)
    if rng.random() < 0.5: cnt = np.maximum(cnt, ci + 3).astype(np.uint32)
    o.build(k, km, cnt)
    want = [dig(o.array_bytes(w, a)) for a in range(nb) for w in ("tag", "value")] + [dig(o.array_bytes("km_back"))]
    for env in [HOOKS[int(rng.integers(0, len(HOOKS)))], HOOKS[int(rng.integers(0, len(HOOKS)))]]:
        for v in VARS: os.environ.pop(v, None)
        os.environ.update(env)
        for r in range(reps):
            m = KModel(ci, cs, nh, nb); m.build_packed(k, km, cnt)
            got = [dig(m.download(w, a)) for a in range(nb) for w in ("tag", "value")] + [dig(m.download("km_back"))]
            builds += 1
            if got != want:
                bad += 1
                print("WRONG", (k, ci, cs, nh, nb, n, seed, env), "build", r, [i for i in range(len(want)) if got[i] != want[i]], flush=True)
            del m
    done += 1
    if done % 10 == 0: print(f"[{time.time()-t0:.0f}s] {done} configurations, {builds} builds, {bad} wrong", flush=True)
print(f"SOAK-STRESS: {done} configurations, {builds} builds, {bad} wrong, {time.time()-t0:.0f}s")
