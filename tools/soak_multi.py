"""Determinism soak of the C++ multi-handle builds (kmx_build_from_kmc_multi_ex): the bench's stream as a KMC1 database, built again and
again by H handles on cuda:0 with every partition / transport; the digest of every array + km_back and the statistics must not
change from build to build, nor between handles, nor between partitions (they all are the reference's model).
usage: python tools/soak_multi.py [n_kmers] [reps]"""
import hashlib, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import write_kmc1_from_device
from kmcex_amd import KModel, api, synth_torch

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
k, ci, cs, nh, nb = 31, 1, 1023, 7, 5
dev = torch.device("cuda", 0)
km, cnt = synth_torch.make_stream(n, k, ci, cs, dev)
tmp = tempfile.mkdtemp(prefix="kmx_soak_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)


def digest(m):
    h = hashlib.sha256()
    for a in range(nb):
        h.update(m.download("tag", a).tobytes()); h.update(m.download("value", a).tobytes())
    h.update(m.download("km_back").tobytes())
    st = m.stats()
    return h.hexdigest()[:16], (st.attempts, st.successes, st.rest_entries)


try:
    db = os.path.join(tmp, "db")
    write_kmc1_from_device(db, km, cnt, k, ci, cs)
    del km, cnt
    ref = None
    t0 = time.time()
    builds = 0
    for part, hs in (("range", 1), ("range", 2), ("range", 3), ("range", 5), ("ring", 2), ("ring", 5), ("range-rccl", 1)):
        ms = [KModel(ci, cs, nh, nb) for _ in range(hs)]
        for r in range(reps):
            api.init_multi(ms, db, part)
            builds += 1
            for j in ((0, hs - 1) if r else range(hs)):          # every handle after the first build, then the first and the last
                d = digest(ms[j])
                ref = ref or d
                assert d == ref, (part, hs, r, j, d, ref)
        for m in ms:
            m.close()
        print(f"[{time.time() - t0:.0f}s] {part}, {hs} handle(s): {reps} builds of {n} k-mers, digest {ref[0]} stats {ref[1]} every time, on every handle checked", flush=True)
    print(f"MULTI SOAK OK: {builds} builds, none different")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
