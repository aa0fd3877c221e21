"""Randomised stress of the position-range partition's kernels (range_kernels.h) against the CPU oracle: ONE rank holds every
list and owns every range (each exchange is a word "sent" to itself), so the emit / verdict + detect / apply / ordered resolve /
commit kernels run alone on configurations biased to heavy contention -- tiny arrays where nearly every candidate is contended
(the resolver iterates many times), claim bins that overflow on the owner, several blocks with a partial last one.
With `world=N` (or `world=rand`) the N ranks are threads of this process (tests/thread_comm.py) that share the GPU: the lists are spread
over the ranks, every rank owns a range of every array, the words really travel between the handles, and EVERY rank's model is
compared with the oracle.
With `cxx` the build goes through the C++ entry instead (kmx_build_from_kmc_multi_ex on a KMC1 database written for the case: world
handles on cuda:0, host threads, the words of a round through the peer-mapped inboxes -- part=range -- or whole arrays -- part=ring);
`msg=counted` asks the Python path for counted messages instead of the fixed-size ones.
usage: python tools/stress_range.py [seconds] [seed] [big] [world=N|world=rand] [part=range|ring] [cxx] [msg=fixed|counted]"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from kmcex_amd import KModel, synth
from kmcex_amd import dist as kd

from thread_comm import run_threads
world_arg = next((a.split("=")[1] for a in sys.argv if a.startswith("world=")), None)
part = next((a.split("=")[1] for a in sys.argv if a.startswith("part=")), "range")
cxx = "cxx" in sys.argv
msg = next((a.split("=")[1] for a in sys.argv if a.startswith("msg=")), None)
if msg:
    os.environ["KMX_RANGE_MESSAGES"] = msg
from kmcex_amd import api, kmcdb
tmpdir = tempfile.mkdtemp(prefix="kmx_stress_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)      # part=ring: the ring of whole arrays through the same harness
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
SIZES = [600000, 1500000, 3000000] if "big" in sys.argv else [40, 300, 3000, 30000, 120000, 300000, 700000]
dev = torch.device("cuda", 0)
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
    total = len(cnt) if rng.random() < 0.5 else None
    # tiny arrays: declare fewer k-mers than are inserted?  No: the sizes must be the sequential build's; contention comes from n << 2^18 lists on arrays sized for n
    o.build(k, km, cnt); so = o.stats()
    W = (k + 31) // 32
    tk = torch.from_numpy(np.ascontiguousarray(km, dtype=np.uint64).view(np.int64).reshape(-1, W) if W > 1 else np.ascontiguousarray(km, dtype=np.uint64).view(np.int64)).to(dev)
    tc = torch.from_numpy(np.ascontiguousarray(cnt, dtype=np.uint32).view(np.int32)).to(dev)
    world = 1 if world_arg is None else (int(rng.integers(2, 17)) if world_arg == "rand" else int(world_arg))
    tag = (k, ci, cs, nh, nb, n, seed, world)
    if cxx:
        db = os.path.join(tmpdir, "db")
        kmcdb.write_kmc1(db, km, cnt, k, ci, cs, total_override=total)
        models = [KModel(ci, cs, nh, nb) for _ in range(world)]
        api.init_multi(models, db, part)
        m = models[0]
        for r, mr in enumerate(models[1:], 1):
            for a in range(nb):
                assert np.array_equal(mr.download("tag", a), m.download("tag", a)) and np.array_equal(mr.download("value", a), m.download("value", a)), ("handle", r, "array", a, tag)
            assert np.array_equal(mr.download("km_back"), m.download("km_back")), ("handle", r, "km_back", tag)
            sr, s0 = mr.stats(), m.stats()
            assert (sr.attempts, sr.successes, sr.rest_entries) == (s0.attempts, s0.successes, s0.rest_entries), ("handle", r, "stats", tag)
            mr.close()
    elif world > 1:
        def rank_body(rank, comm):
            lo, hi = kd.split_batch(len(cnt), world, rank)
            mr = KModel(ci, cs, nh, nb)
            kd.build_sharded(kd.DeviceEngine(mr, dev), comm, k, nb, 1 if ci == 1 else 3, tk[lo:hi].contiguous(), tc[lo:hi].contiguous(), partition=part)
            return mr
        models = run_threads(world, rank_body)
        m = models[0]
        for r, mr in enumerate(models[1:], 1):                  # every rank must hold the model rank 0 holds (compared with the oracle below)
            for a in range(nb):
                assert np.array_equal(mr.download("tag", a), m.download("tag", a)) and np.array_equal(mr.download("value", a), m.download("value", a)), ("rank", r, "array", a, tag)
            assert np.array_equal(mr.download("km_back"), m.download("km_back")), ("rank", r, "km_back", tag)
            sr, s0 = mr.stats(), m.stats()
            assert (sr.attempts, sr.successes, sr.rest_entries) == (s0.attempts, s0.successes, s0.rest_entries), ("rank", r, "stats", tag)
            mr.close()
    else:
        m = KModel(ci, cs, nh, nb)
        kd.build_sharded(kd.DeviceEngine(m, dev), kd.Comm(), k, nb, 1 if ci == 1 else 3, tk, tc, partition=part)
    st = m.stats()
    for a in range(nb):
        assert np.array_equal(m.download("tag", a), o.array_bytes("tag", a)), ("tag", a, tag)
        assert np.array_equal(m.download("value", a), o.array_bytes("value", a)), ("value", a, tag)
    assert np.array_equal(m.download("km_back"), o.array_bytes("km_back")), ("km_back", tag)
    for f in range(st.bf_num):
        assert np.array_equal(m.download("bf", f), o.array_bytes("bf", f)) and np.array_equal(m.download("bf_back", f), o.array_bytes("bf_back", f)), ("bloom", f, tag)
    assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries), ("stats", tag, (st.attempts, st.successes, st.rest_entries), (so.attempts, so.successes, so.rest_entries))
    q = np.concatenate([km.reshape(-1, W)[:: max(1, len(cnt) // 2000)].reshape(-1), synth.random_kmers(500, k, seed_k=seed + 7).reshape(-1)])
    assert np.array_equal(m.kmer_to_occ_packed(q), o.query_packed(k, q, threads=8)), ("query", tag)
    contended += st.contended
    m.close(); del m
    done += 1
    if done % (5 if "big" in sys.argv else 20) == 0: print(f"[{time.time()-t0:.0f}s] {done} configurations, {contended} contended k-mers decided in list order, all bit-exact", flush=True)
shutil.rmtree(tmpdir, ignore_errors=True)
print(f"{part.upper()}{' (C++ entry)' if cxx else ''} STRESS OK (world {world_arg or 1}): {done} configurations, {contended} contended k-mers, {time.time()-t0:.0f}s")
