"""Re-run ONE configuration of tools/stress_parity.py under several hook settings and report which arrays differ from the oracle.
usage: python tools/repro_case.py k ci cs nh nb n seed"""
import os as _os; _os.environ.setdefault("KMX_TEST_HOOKS", "1")   # forced code paths are test hooks
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from kmcex_amd import KModel, synth
k, ci, cs, nh, nb, n, seed = (int(x) for x in sys.argv[1:8])
km0, cnt0 = synth.make_stream(n, k, ci, cs, seed_k=seed, seed_c=seed + 1)
for forced in (0, 1):
    cnt = np.maximum(cnt0, ci + 3).astype(np.uint32) if forced else cnt0
    o = O.OracleModel(ci, cs, nh, nb); o.build(k, km0, cnt); so = o.stats()
    for env in (dict(), dict(KMX_NSUB0="0", KMX_NSUB1="0"), dict(KMX_NSUB0="0", KMX_NSUB1="0", KMX_PIPE="0"), dict(KMX_NSUB0="0", KMX_NSUB1="0", KMX_KMB_HOST="0"),
                dict(KMX_NSUB0="0", KMX_NSUB1="0", KMX_PIPE="0", KMX_KMB_HOST="0"), dict(KMX_NSUB0="1", KMX_NSUB1="1"), dict(KMX_FIN_GLOBAL="1")):
        for rep in range(3):
            for v in ("KMX_NSUB0", "KMX_NSUB1", "KMX_FIN_GLOBAL", "KMX_RESOLVE_GATHER", "KMX_KMB_DIRECT", "KMX_PIPE", "KMX_KMB_HOST"): os.environ.pop(v, None)
            os.environ.update(env)
            m = KModel(ci, cs, nh, nb); m.build_packed(k, km0, cnt); st = m.stats()
            bad = []
            for a in range(nb):
                for w in ("tag", "value"):
                    d = np.unpackbits(m.download(w, a) ^ o.array_bytes(w, a)).sum()
                    if d: bad.append((w, a, int(d)))
            if not np.array_equal(m.download("km_back"), o.array_bytes("km_back")): bad.append("km_back")
            if (st.attempts, st.successes, st.rest_entries) != (so.attempts, so.successes, so.rest_entries): bad.append(("stats", st.attempts, st.successes, st.rest_entries, so.attempts, so.successes, so.rest_entries))
            print(f"forced={forced} env={env} rep={rep} contended={st.contended} fin_iters={st.finisher_iters} -> {'OK' if not bad else bad}", flush=True)
            del m
