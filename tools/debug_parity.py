import os as _os; _os.environ.setdefault("KMX_TEST_HOOKS", "1")   # forced code paths are test hooks
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from kmcex_amd import KModel, api, synth

k = 31
km = synth.random_kmers(1000, k, seed_k=5)
strs = synth.to_strings(km, k)
seeds = [O.lib().kmo_hash_seed(i) for i in (0, 1, 34)]
for whole in (True, False):
    dev = api.debug_hash(k, km, seeds, whole)
    exp = np.array([[O.murmur64((s if whole else s[1:-1]).encode(), si) for si in (0, 1, 34)] for s in strs], dtype=np.uint64)
    print("hash whole=%d match: %d / %d" % (whole, int((dev == exp).sum()), exp.size), flush=True)
    if not (dev == exp).all():
        print(strs[0], hex(int(dev[0, 0])), hex(int(exp[0, 0])))
dev = api.debug_min_kmer(k, km)
exp = synth.from_strings([O.min_kmer(s) for s in strs], k)
print("min_kmer match:", int((dev == exp).sum()), len(exp), flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
ci, cs, nh, nb = 1, 1023, 7, 5
km, cnt = synth.make_stream(n, k, ci, cs)
m = KModel(ci, cs, nh, nb); m.build_packed(k, km, cnt)
o = O.OracleModel(ci, cs, nh, nb); o.build(k, km, cnt)
st, so = m.stats(), o.stats()
for f in ("n_total", "n_km", "attempts", "successes", "rest_entries", "km_byte_size", "byte_km_back"):
    print(f, getattr(st, f), getattr(so, f))
print("fast", st.fast_commits, "contended", st.contended, "fin_iters", st.finisher_iters, "blocks", st.blocks, "rounds", st.rounds)
for name, idxs in (("bf", [0]), ("bf_back", [0]), ("km_back", [0]), ("tag", range(nb)), ("value", range(nb)), ("claims", range(nb))):
    for i in idxs:
        a = m.download(name, i)
        if name == "claims":
            print(name, i, "nonzero bytes", int((a != 0).sum())); continue
        b = o.array_bytes(name, i)
        x = np.unpackbits(a ^ b)
        print(name, i, "len", len(a), len(b), "diff bits", int(x.sum()), "popcount dev/ora", int(np.unpackbits(a).sum()), int(np.unpackbits(b).sum()), flush=True)
q = np.concatenate([km[::7], synth.random_kmers(5000, k, seed_k=0xABCDEF0123)])
r1, r2 = m.kmer_to_occ_packed(q), o.query_packed(k, q)
print("query mismatches", int((r1 != r2).sum()), len(q))
