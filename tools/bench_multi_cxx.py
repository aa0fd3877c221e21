"""KModel::init(db) by several handles from C++ (kmx_build_from_kmc_multi_ex), timed: ring and range partition, 1 .. H handles
that share cuda:0, on the bench's 10^8-k-mer stream written as a KMC1 database in tmpfs.  Beside it the Python-driven range
partition (dist.build_range_sharded: the caller moves the words, one host wait per round) on the same stream, one rank.
With KMX_INIT_TRACE=1 the library prints the phase times of handle 0 (the rounds alone).
usage: python tools/bench_multi_cxx.py [n_kmers] [max_handles] [reps] [parts=ring,range] [min_handles=1] [nopython]"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import write_kmc1_from_device
from kmcex_amd import KModel, api, synth_torch
from kmcex_amd import dist as kd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
parts = next((a.split("=")[1].split(",") for a in sys.argv if a.startswith("parts=")), ["ring", "range"])
h_min = int(next((a.split("=")[1] for a in sys.argv if a.startswith("min_handles=")), 1))
k, ci, cs, nh, nb = 31, 1, 1023, 7, 5
dev = torch.device("cuda", 0)
km, cnt = synth_torch.make_stream(n, k, ci, cs, dev)
tmp = tempfile.mkdtemp(prefix="kmx_multi_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
out = {"n_kmers": int(cnt.numel()), "k": k, "nh": nh, "nb": nb}
try:
    db = os.path.join(tmp, "db")
    write_kmc1_from_device(db, km, cnt, k, ci, cs)
    ref = None
    for part in parts:
        for h in range(h_min, H + 1):
            ms = [KModel(ci, cs, nh, nb) for _ in range(h)]
            ts = []
            for r in range(reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                api.init_multi(ms, db, part)
                ts.append(time.perf_counter() - t0)
            st = ms[0].stats()
            sig = (st.attempts, st.successes, st.rest_entries)
            ref = ref or sig
            assert sig == ref, (part, h, sig, ref)
            out[f"{part}_{h}_handles_ms"] = [round(t * 1e3, 2) for t in ts[1:]]
            print(part, h, "handles:", out[f"{part}_{h}_handles_ms"], "ms per init(db) of", cnt.numel(), "k-mers", flush=True)
            for m in ms:
                m.close()
    if "nopython" in sys.argv:
        print(json.dumps(out))
        sys.exit(0)
    # the Python-driven range partition on one rank (device-resident listing: no decode, no routing)
    m = KModel(ci, cs, nh, nb)
    eng = kd.DeviceEngine(m, dev)
    ts = []
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kd.build_sharded(eng, kd.Comm(), k, nb, 1, km, cnt, partition="range")
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    st = m.stats()
    assert ref is None or (st.attempts, st.successes, st.rest_entries) == ref
    out["python_range_1_rank_ms"] = [round(t * 1e3, 2) for t in ts[1:]]
    print("python range, 1 rank:", out["python_range_1_rank_ms"], flush=True)
    m.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
print(json.dumps(out))
