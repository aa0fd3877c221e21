#!/usr/bin/env python3
"""bench.py -- KModel insert + query throughput on MI355X (BASELINE.json metric), one JSON line on stdout.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic input: one full KModel build
(kmx_build_dev = pass-1 histogram + classification/Bloom insert + ordered coupled-array insert + rest build)
of the configured stream, already resident in HBM.  Query steps (one batched kmer_to_occ over the query set)
are timed in a second, separately bracketed region.  `value` is the insert rate (k-mers encoded per second,
whole job); the query rate is reported beside it.

Workload at N=1: BASELINE.json configs[1] -- synthetic 100 M distinct canonical 31-mers, nh=7 nb=5 ci=1
cs=1023, D1 counts.  N>1: `value` = ONE model (kmodel.hpp:57-86 is one model) over the concatenation of
the N ranks' streams, built by the N ranks together (kmcex_amd/dist.py: k-mer routing all-to-all, ring of arrays over
send/recv, OR-merged filters, replica queries) -- weak scaling, N * kmers_per_gpu k-mers in
the model; `replica_value` beside it = N independent models, one per rank (no data-path collective).  At N=1 the two
coincide and `value` is the plain build; `single_model` keeps the ring code's N=1 figure.  `init_db` (N=1) times the
reference's real entry point, KModel::init on a KMC database of the same stream in tmpfs (kmodel.hpp:57-86), host feed
included.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

G = 32  # bytes per touched position (HBM access granule)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--kmers", dest="n", type=int, default=100_000_000, help="k-mer draws per rank")
    ap.add_argument("--kmer-len", dest="k", type=int, default=31)
    ap.add_argument("--ci", type=int, default=1)
    ap.add_argument("--cs", type=int, default=1023)
    ap.add_argument("--nh", type=int, default=7)
    ap.add_argument("--nb", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="k-mers of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-init-db", action="store_true", help="skip the KModel::init(database) leg")
    ap.add_argument("--no-query-strings", action="store_true", help="skip the kmer_to_occ(vector<string>) leg")
    ap.add_argument("--genome-bases", type=int, default=30_000_000, help="bases of the genome-like stream of the query_genome leg (0 = skip)")
    ap.add_argument("--no-single-model", action="store_true", help="skip the one-model-over-all-ranks leg")
    ap.add_argument("--single-model-steps", type=int, default=0, help="0: --steps when N > 1 (it is the headline there), 2 at N = 1")
    ap.add_argument("--no-other-partition", action="store_true", help="N > 1: do not also time the partition that was not selected")
    ap.add_argument("--no-cxx-multi", action="store_true", help="N > 1: skip the C++ multi-GPU entry (kmx_build_from_kmc_multi_ex driven from ONE process, a child of rank 0)")
    ap.add_argument("--cxx-multi-child", type=str, default="", help=argparse.SUPPRESS)     # internal: run the C++ multi-GPU leg on these devices ("0,1,2,...") and print its JSON
    ap.add_argument("--partition", choices=("ring", "range"), default="ring",
                    help="how ONE model is spread over the ranks: ring = arrays owned whole, lists travel (send/recv); range = every array cut by "
                         "position range, commits + triples out and verdicts back by all-to-all (the north star's partition, SURVEY.md 8e(1))")
    return ap.parse_args()


def write_kmc1_from_device(prefix, km, cnt, k, ci, cs):
    """The listing as a KMC1 database -- records packed on the GPU, written once.  Plumbing."""
    import struct
    from kmcex_amd import kmcdb
    p = kmcdb.lut_prefix_len(k)
    sb = (k - p) // 4                                             # suffix bytes per record
    csz = 1
    while cs >= (1 << (8 * csz)):
        csz += 1
    n = km.numel()
    with open(prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        step = 1 << 25
        for lo in range(0, n, step):
            x, c = km[lo:lo + step], cnt[lo:lo + step].to(torch.int64)
            cols = [((x >> (8 * (sb - 1 - j))) & 0xFF).to(torch.uint8) for j in range(sb)] + [((c >> (8 * b)) & 0xFF).to(torch.uint8) for b in range(csz)]
            f.write(torch.stack(cols, dim=1).cpu().numpy().tobytes())
        f.write(b"KMCS")
    pre = km >> (2 * (k - p))
    lut = torch.searchsorted(pre, torch.arange(4 ** p, dtype=torch.int64, device=km.device), right=False).cpu().numpy().astype(np.uint64)
    hdr = struct.pack("<IIIIIIQB3xI", k, 0, csz, p, ci, cs & 0xFFFFFFFF, n, 0, cs >> 32)
    hdr = hdr + b"\0" * (64 - len(hdr))
    with open(prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        f.write(hdr)
        f.write(struct.pack("<I", 64))
        f.write(b"KMCP")


def write_kmc2_from_device(prefix, km, cnt, k, ci, cs, n_bins=512, signature_len=5):
    """The same listing in the layout KMC 3 itself writes (version 0x200, kmc_file.cpp:188-235): records bin-major, one LUT per
    bin, sorted only inside a bin -- the listing (and so the insert order) is not globally sorted.  The bin of a k-mer is a hash
    of it (the listing reader never looks at signatures; kmcex_amd/kmcdb.py write_kmc2 is the numpy twin).  Returns the listing order."""
    import struct
    from kmcex_amd import kmcdb
    p = kmcdb.lut_prefix_len(k)
    sb = (k - p) // 4
    csz = 1
    while cs >= (1 << (8 * csz)):
        csz += 1
    n = km.numel()
    h = (km * (-7046029254386353131)) >> 40                       # 0x9E3779B97F4A7C15 as int64: same bits as the uint64 product
    bins = (h & ((1 << 24) - 1)) % n_bins
    order = torch.sort(bins, stable=True).indices                 # the input is sorted, so every bin stays sorted
    kmo, cno, bo = km[order], cnt[order], bins[order]
    del h, bins
    with open(prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        step = 1 << 25
        for lo in range(0, n, step):
            x, c = kmo[lo:lo + step], cno[lo:lo + step].to(torch.int64)
            cols = [((x >> (8 * (sb - 1 - j))) & 0xFF).to(torch.uint8) for j in range(sb)] + [((c >> (8 * b)) & 0xFF).to(torch.uint8) for b in range(csz)]
            f.write(torch.stack(cols, dim=1).cpu().numpy().tobytes())
        f.write(b"KMCS")
    key = bo * (4 ** p) + (kmo >> (2 * (k - p)))                  # (bin, prefix) ascending along the file
    lut = torch.searchsorted(key, torch.arange(n_bins * 4 ** p, dtype=torch.int64, device=km.device), right=False).cpu().numpy().astype(np.uint64)
    sig_map = np.zeros(4 ** signature_len + 1, dtype=np.uint32)
    hdr = struct.pack("<IIIIIIIQB", k, 0, csz, p, signature_len, ci, cs & 0xFFFFFFFF, n, 0)
    hdr = hdr + b"\0" * (60 - len(hdr)) + struct.pack("<I", 0x200)
    with open(prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        f.write(struct.pack("<Q", n))
        f.write(sig_map.tobytes())
        f.write(hdr)
        f.write(struct.pack("<I", 64))
        f.write(b"KMCP")
    return order


def init_db_leg(a, km, cnt, reps=3, layout="kmc1"):
    """KModel::init(db_file) end to end (kmodel.hpp:57-86): listing decode on the host cores, pinned H2D, insert, rest build.
    layout "kmc1": one sorted listing (KMC 1 / the reference's test databases); "kmc2": what KMC 3 writes -- 512 bins, each
    sorted, listed bin by bin (kmc_file.cpp:188-235): another insert order, another (equally valid) model."""
    import shutil
    from kmcex_amd import KModel
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="kmx_bench_db_", dir=base)
    tag = "init_db" if layout == "kmc1" else "init_db_kmc2"
    try:
        db = os.path.join(tmp, "db")
        if layout == "kmc1":
            write_kmc1_from_device(db, km, cnt, a.k, a.ci, a.cs)
        else:
            write_kmc2_from_device(db, km, cnt, a.k, a.ci, a.cs)
        m = KModel(a.ci, a.cs, a.nh, a.nb)
        ts = []
        for _ in range(reps + 1):                                 # first call allocates: warm-up
            t0 = time.perf_counter()
            m.init(db)
            ts.append(time.perf_counter() - t0)
        st = m.stats()
        m.close()
        best, mean = min(ts[1:]), sum(ts[1:]) / reps
        multi = {}
        if layout == "kmc1":
            # the same database through the C++ multi-GPU entry with the ONE handle this box has: the north star's position-range
            # partition alone (every word "sent" to itself) -- through the peer-mapped inboxes and through RCCL messages
            from kmcex_amd import api
            for part, key in (("range", "init_db_range_cxx"), ("range-rccl", "init_db_range_rccl_cxx")):
                try:
                    mm = KModel(a.ci, a.cs, a.nh, a.nb)
                    tt = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        api.init_multi([mm], db, part)
                        tt.append(time.perf_counter() - t0)
                    ok = mm.stats().attempts == st.attempts
                    mm.close()
                    multi[f"{key}_1_handle_ms"] = min(tt[1:]) * 1e3
                    multi[f"{key}_1_handle_value"] = km.numel() / min(tt[1:])
                    multi[f"{key}_same_attempts"] = ok
                except Exception as e:  # noqa: BLE001
                    multi[f"{key}_error"] = repr(e)
            multi["init_db_range_cxx_what"] = ("kmx_build_from_kmc_multi_ex(KMX_PARTITION_RANGE / _RANGE_RCCL) with one handle: KModel::init(db) through the range "
                                               "partition's kernels and transports alone; N handles need N GPUs (tests run 1-8 handles on one)")
        what = ("KModel::init(database in tmpfs): KMC listing decode + pinned hipMemcpyAsync + insert + rest build, wall clock" if layout == "kmc1" else
                "the same on a KMC2-layout database (what KMC 3 writes): 512 bins listed bin by bin, unsorted insert order")
        return {f"{tag}_value": km.numel() / mean, f"{tag}_ms": mean * 1e3, f"{tag}_best_ms": best * 1e3, f"{tag}_reps": reps,
                f"{tag}_what": what, f"{tag}_attempts": st.attempts, f"{tag}_bytes": os.path.getsize(db + ".kmc_suf"), **multi}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def genome_leg(a, dev, n_bases, reps=2):
    """Throughput where the data is NOT uniformly random: all overlapping k-mers of a random sequence (their de Bruijn
    neighbours are in the set, which is what the query's neighbour disambiguation, kmodel.hpp:286-359, feeds on).  Insert,
    query of the stored k-mers, query of their successors (3 of 4 absent), and -- from one accounting pass, never timed --
    the fraction of queries that entered the neighbour path."""
    from kmcex_amd import KModel, synth_torch
    k = a.k
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    bases = torch.randint(0, 4, (n_bases,), dtype=torch.int64, device=dev, generator=g)
    n = n_bases - k + 1
    v = torch.zeros(n, dtype=torch.int64, device=dev)
    for j in range(k):
        v = (v << 2) | bases[j:j + n]
    v &= (1 << (2 * k)) - 1
    del bases
    km = torch.unique(torch.minimum(v, synth_torch.revcomp(v, k)), sorted=True)
    del v
    cnt = synth_torch.d1_counts(km.numel(), a.ci, a.cs, 2, dev)
    n = km.numel()
    m = KModel(a.ci, a.cs, a.nh, a.nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    succ = ((km[: n // 2] << 2) | 1) & ((1 << (2 * k)) - 1)        # successors of stored k-mers: 1 of 4 is in the sequence
    out = torch.empty(n, dtype=torch.int32, device=dev)

    def timed(fn):
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sum(ts[1:]) / reps
    t_ins = timed(lambda: m.build_dev(k, km.data_ptr(), cnt.data_ptr(), n))
    t_q = timed(lambda: m.kmer_to_occ_dev(km.data_ptr(), n, out.data_ptr()))
    t_s = timed(lambda: m.kmer_to_occ_dev(succ.data_ptr(), succ.numel(), out.data_ptr()))
    nonzero = float((out[: succ.numel()] != 0).float().mean().item())
    m.set_profile(2)                                              # accounting variant of the query kernel
    m.kmer_to_occ_dev(km.data_ptr(), n, out.data_ptr())
    f_stored = m.stats().query_neighbour_calls / n
    m.kmer_to_occ_dev(succ.data_ptr(), succ.numel(), out.data_ptr())
    f_succ = m.stats().query_neighbour_calls / succ.numel()
    m.set_profile(False)
    st = m.stats()
    m.close()
    return {"genome_bases": n_bases, "genome_kmers": n, "genome_insert_value": n / t_ins, "genome_insert_ms": t_ins * 1e3,
            "query_genome_value": n / t_q, "query_genome_ms": t_q * 1e3,
            "query_genome_successors_value": succ.numel() / t_s, "query_genome_successors_answered_nonzero": nonzero,
            "query_genome_neighbour_path_fraction": {"stored": f_stored, "successors": f_succ},
            "genome_attempts_per_kmer": st.attempts / max(st.n_km, 1),
            "genome_what": f"all overlapping {k}-mers of a random {n_bases}-base sequence, canonical, distinct, D1 counts: insert (kmx_build_dev), "
                           "kmer_to_occ of the stored k-mers and of their successors, resident in HBM; neighbour_path_fraction from an accounting pass"}


def cxx_multi_leg(a, km, cnt, world, reps=3, devices=None):
    """KModel::init(db) by `world` handles, one per GPU, from THIS process (include/kmx.h kmx_build_from_kmc_multi_ex; what
    KMX_DEVICES=0,1,... KMX_PARTITION=... gives a caller of the reference's API): ms per build of rank 0's stream."""
    import shutil
    from kmcex_amd import KModel, api
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="kmx_bench_multi_", dir=base)
    out = {"kmers": int(km.numel()), "handles": world, "devices": devices or list(range(world)), "scaling": "strong (one stream of kmers_per_gpu k-mers built by all GPUs)",
           "what": "kmx_build_from_kmc_multi_ex from one process, one handle per GPU: wall clock of KModel::init(database in tmpfs), best of the timed builds"}
    try:
        db = os.path.join(tmp, "db")
        write_kmc1_from_device(db, km, cnt, a.k, a.ci, a.cs)
        ref = None
        for part in ("range", "range-rccl", "ring"):
            try:
                ms = [KModel(a.ci, a.cs, a.nh, a.nb, device=d) for d in (devices or list(range(world)))]
                ts = []
                for _ in range(reps + 1):
                    t0 = time.perf_counter()
                    api.init_multi(ms, db, part)
                    ts.append(time.perf_counter() - t0)
                st = ms[-1].stats()
                sig = (st.attempts, st.successes, st.rest_entries)
                ref = ref or sig
                out[part] = {"ms_per_build": min(ts[1:]) * 1e3, "value": km.numel() / min(ts[1:]), "all_ms": [t * 1e3 for t in ts[1:]], "same_model": sig == ref}
                for m in ms:
                    m.close()
            except Exception as e:  # noqa: BLE001
                out[part] = {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def source_digest():
    """sha256 over the library's sources: the PMC file carries the digest of the code its counters were taken on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "kmcex_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "kmx.h"), "rb").read())
    return h.hexdigest()[:16]


def query_strings_leg(a, m, q, out, reps=3):
    """The reference's only batch query API, kmer_to_occ(vector<string>) (kmodel.hpp:90-98), end to end: n separate host
    strings in, answers in a host array out (kmx_query_strings: worker threads pack chunks into pinned slots, hipMemcpyAsync
    both ways under the kernel of the chunk before).  Wall clock; the strings are built once, outside the timed region."""
    import ctypes as C
    k, nq = a.k, q.numel()
    rows = torch.zeros((nq, 32), dtype=torch.uint8, device=q.device)     # 31 characters + NUL: what a std::string's buffer holds
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=q.device)
    for pos in range(k):
        rows[:, pos] = lut[((q >> (2 * (k - 1 - pos))) & 3)]
    host = rows.cpu().numpy()
    del rows
    ptrs = (host.ctypes.data + np.arange(nq, dtype=np.uint64) * np.uint64(32)).astype(np.uint64)
    ans = np.zeros(nq, dtype=np.int32)
    ts = []
    for _ in range(reps + 1):                                            # first call allocates the pinned slots: warm-up
        t0 = time.perf_counter()
        rc = m.L.kmx_query_strings(m.h, C.cast(ptrs.ctypes.data, C.POINTER(C.c_char_p)), k, nq, ans.ctypes.data)
        ts.append(time.perf_counter() - t0)
        if rc != 0:
            raise RuntimeError(f"kmx_query_strings: {rc}")
    same = bool(np.array_equal(ans, out.cpu().numpy()))                 # against the packed device-resident query of the same k-mers
    mean = sum(ts[1:]) / reps
    return {"query_strings_value": nq / mean, "query_strings_ms": mean * 1e3, "query_strings_best_ms": min(ts[1:]) * 1e3,
            "query_strings_reps": reps, "query_strings_equal_packed_answers": same, "query_strings_host_cpus": len(os.sched_getaffinity(0)),
            "query_strings_what": f"kmx_query_strings = kmer_to_occ(vector<string>): {nq} separate host strings in, host int32 out, wall clock"}


def sync_all(distributed):
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()


def cpu_baseline(a):
    """The real reference (oracle/_ref, kind=reference) or the CPU oracle (kind=port) on a bounded sample of the
    same workload, timed on this host's cores.  Never on the product path; reported, not a target."""
    import oracle_lib as O
    from common import query_set
    from kmcex_amd import kmcdb, synth
    n = a.cpu_sample
    km, cnt = synth.make_stream(n, a.k, a.ci, a.cs)
    q = query_set(km, a.k, max_present=min(len(km), 1_000_000))      # +10 % absent draws
    ncpu = os.cpu_count() or 1
    out = {"unit": "k-mers/s", "sample": f"{len(cnt)} synthetic {a.k}-mers (same generator, D1 counts), "
           f"insert = KModel::init incl. both passes and rest build; query = {len(q)} k-mers"}
    with tempfile.TemporaryDirectory(prefix="kmx_cpu_") as tmp:
        if O.have_ref():
            db = os.path.join(tmp, "db")
            kmcdb.write_kmc1(db, km, cnt, a.k, a.ci, a.cs)
            t_ins = O.ref_build(db, db + ".m", a.ci, a.cs, a.nh, a.nb)
            O.ref_query(db + ".m", synth.to_strings(q, a.k), db, t_num=ncpu)
            t_q = O.last_ref_query_seconds
            out.update(kind="reference", cores=a.nb, query_cores=ncpu)
        else:
            m = O.OracleModel(a.ci, a.cs, a.nh, a.nb)
            t0 = time.time()
            m.build(a.k, km, cnt)
            t_ins = time.time() - t0
            t0 = time.time()
            m.query_packed(a.k, q, threads=ncpu)
            t_q = time.time() - t0
            out.update(kind="port", cores=a.nb, query_cores=ncpu)
    out["value"] = len(cnt) / t_ins
    out["query_value"] = len(q) / t_q
    out["host_cpus"] = ncpu
    return out


def headline(a, world, n, q, n_all, nq_all, t_ins, t_q, st, roof, cpu, extra, init_db):
    line = {
        "metric": "k-mers/s encoded (insert) + k-mers/s queried, k=31 nh=7; % HBM-BW roofline",
        "value_is": "k-mers/s encoded (insert); the query rate is query_value",
        "value": n_all * a.steps / t_ins, "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": t_ins / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"synthetic {n} distinct canonical {a.k}-mers per GPU, D1 counts, insert then query "
                               f"({q.numel()} queries: all inserted k-mers shuffled, half reverse-complemented, +10% absent)",
                   "k": a.k, "nh": a.nh, "nb": a.nb, "ci": a.ci, "cs": a.cs, "kmers_per_gpu": n,
                   "parallelism": f"{world} independent model(s), one per GPU (value); one model over all ranks in single_model"},
        "query_value": nq_all * a.steps / t_q, "query_ms_per_step": t_q / a.steps * 1e3,
        "stats": {"n_km": st.n_km, "n_bf": list(st.n_bf)[: st.bf_num], "attempts": st.attempts, "successes": st.successes,
                  "rest_entries": st.rest_entries, "fast_commits": st.fast_commits, "contended": st.contended,
                  "finisher_iters": st.finisher_iters, "blocks": st.blocks, "rounds": st.rounds},
        "roofline": roof, "cpu_baseline": cpu,
    }
    line.update(extra)
    line.update(init_db)
    return line


def promote_single_model(line, single, world):
    """N > 1: the headline is the ONE model the N ranks build together; the N independent models become `replica_value`."""
    line["single_model"] = single
    if world == 1 or single is None:                             # N = 1, or the leg was switched off (--no-single-model)
        return line
    if "value" not in single:
        # the leg failed: the headline (ONE model) was NOT measured -- the replicas' rate must not stand in for it
        line["replica_value"], line["replica_ms_per_step"], line["replica_query_value"] = line["value"], line["ms_per_step"], line["query_value"]
        line["value"], line["ms_per_step"], line["query_value"] = None, None, None
        line["value_is"] = "NOT MEASURED: the single-model leg failed (single_model.error); replica_value = N independent models"
        line["failed"] = True
        return line
    line["replica_value"], line["replica_ms_per_step"], line["replica_query_value"] = line["value"], line["ms_per_step"], line["query_value"]
    line["value"], line["ms_per_step"], line["query_value"] = single["value"], single["ms_per_build"], single["query_value"]
    line["replica_query_ms_per_step"], line["query_ms_per_step"] = line["query_ms_per_step"], single["query_ms_per_step"]
    line["steps"] = single["steps"]
    line["value_is"] = "k-mers/s encoded into ONE model by all ranks together (single_model); replica_value = N independent models"
    how = (f"ring of whole arrays (min({world}, nb) array owners)" if single.get("partition", "ring") == "ring" else
           f"every array cut by position range over the {world} ranks, two all-to-alls per round (commits + triples out, verdicts back)")
    line["config"]["parallelism"] = (f"one model over {world} ranks [--partition {single.get('partition', 'ring')}]: routing all-to-all + {how} + "
                                     f"OR-merged filters + replica queries; replica_value = {world} independent models")
    line["scaling"] = "weak (one model: the ordered chain runs once per round whatever N is -- sub-linear by construction; replica_value scales with N)"
    return line


def single_model_leg(a, m, km, cnt, q, out, rank, world, dev, rehearsal, distributed):
    """ONE model over the concatenation of all ranks' streams (rank order = listing order), built by the ranks together
    and queried over replicas.  Weak scaling: world * kmers_per_gpu k-mers in one model."""
    from kmcex_amd import dist as kd
    comm = kd.Comm()
    eng = kd.DeviceEngine(m, dev)
    bf_num = 1 if a.ci == 1 else 3
    for _ in range(max(1, a.warmup if world > 1 else 1)):
        info = kd.build_sharded(eng, comm, a.k, a.nb, bf_num, km, cnt, partition=a.partition)      # warm-up (allocations)
    sync_all(distributed)
    t0 = time.perf_counter()
    for _ in range(a.single_model_steps):
        info = kd.build_sharded(eng, comm, a.k, a.nb, bf_num, km, cnt, partition=a.partition)
    sync_all(distributed)
    t_b = time.perf_counter() - t0
    st = m.stats()
    # the batch = every rank's query set; each rank answers its own slice against its replica (no collective on the data path)
    t0 = time.perf_counter()
    for _ in range(a.single_model_steps):
        m.kmer_to_occ_dev(q.data_ptr(), q.numel(), out.data_ptr())
    sync_all(distributed)
    t_qq = time.perf_counter() - t0
    owns = 1 if (a.partition == "range" or info.get("arrays_owned")) else 0
    (t_b, t_qq), (n_all, nq_all, sent, working) = kd.reduce_job([t_b, t_qq], [km.numel(), q.numel(), info["bytes_sent"], owns], device="cpu" if rehearsal else dev)
    layout = ({"partition": "range", "cells_owned_rank0": info.get("cells_owned"), "all_to_alls_per_build": info.get("collectives"),
               "messages": info.get("messages"), "fixed_messages_overflowed": bool(info.get("fixed_messages_overflowed")),
               "host_waits_per_round": 0 if info.get("messages") == "fixed" else 1,
               "ranks_holding_lists": min(world, a.nb)} if a.partition == "range" else
              {"partition": "ring", "arrays_owned_rank0": info.get("arrays_owned"), "array_owners": min(world, a.nb),
               "ring_hops_per_build": info["blocks"] * a.nb * sum(1 for x in range(a.nb) if kd.owner_of_array(x, a.nb, world) != kd.owner_of_array((x + 1) % a.nb, a.nb, world))})
    return {"what": ("ONE model over all ranks' streams: routing all-to-all + ring of arrays (send/recv) + OR-merged filters + array broadcast; queries over replicas"
                     if a.partition == "ring" else
                     "ONE model over all ranks' streams: routing all-to-all + arrays cut by position range (commits + triples out, verdicts back: two all-to-alls per round) + "
                     "OR-merged filters + all-gather of the ranges; queries over replicas"),
            **layout, "ranks_doing_ordered_work": working, "idle_ranks_in_the_ordered_rounds": world - working,
            "transport": "gloo, ranks sharing one GPU (rehearsal: rates are not xGMI rates)" if rehearsal else ("nccl (RCCL)" if world > 1 else "none (one rank)"),
            "value": n_all * a.single_model_steps / t_b, "unit": "k-mers/s", "ms_per_build": t_b / a.single_model_steps * 1e3, "kmers": n_all,
            "query_value": nq_all * a.single_model_steps / t_qq, "query_ms_per_step": t_qq / a.single_model_steps * 1e3, "steps": a.single_model_steps, "scaling": "weak",
            "bytes_exchanged_per_build": sent, "blocks": info["blocks"],
            "stats": {"n_km": st.n_km, "attempts": st.attempts, "successes": st.successes, "rest_entries": st.rest_entries}}


_REAL_STDOUT = None


def emit_line(line):
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def cxx_multi_child(a):
    """`bench.py --cxx-multi-child 0,1,...`: the C++ multi-GPU leg in a process of its own (see main): rank 0's stream, regenerated."""
    from kmcex_amd import dist as kd
    from kmcex_amd import synth_torch
    devs = [int(x) for x in a.cxx_multi_child.split(",")]
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)                                                  # (RCCL announces itself on stdout)
    torch.cuda.set_device(devs[0])
    seed_k, seed_c = kd.stream_seeds(0)
    km, cnt = synth_torch.make_stream(a.n, a.k, a.ci, a.cs, torch.device("cuda", devs[0]), seed_k=seed_k, seed_c=seed_c)
    out = cxx_multi_leg(a, km, cnt, len(devs), devices=devs)
    os.write(real, (json.dumps(out) + "\n").encode())


def main():
    a = parse()
    if a.cxx_multi_child:
        cxx_multi_child(a)
        return
    if a.single_model_steps <= 0:
        a.single_model_steps = a.steps if a.gpus > 1 else 2
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare: run the N-rank job as a child (nothing has touched the GPU in this process) and pass on its status
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    # ONE JSON line on stdout, whatever the libraries underneath print there (RCCL announces its version on stdout when a
    # communicator is made): file descriptor 1 becomes stderr for the run, the line goes to the real stdout at the end
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    from kmcex_amd import dist as kd
    rank, local, world = kd.env_world()
    distributed = world > 1
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the KModel hot path has no CPU fallback")
    # rehearsal on a one-GPU box: KMX_BENCH_REHEARSAL=1 puts every rank on cuda:0 and reduces over gloo (CPU tensors);
    # the real multi-GPU run is one rank per GPU over RCCL ("nccl")
    rehearsal = os.environ.get("KMX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    from kmcex_amd import KModel, api, synth_torch

    # ---- synthetic listing of this rank, resident in HBM
    seed_k, seed_c = kd.stream_seeds(rank)
    km, cnt = synth_torch.make_stream(a.n, a.k, a.ci, a.cs, dev, seed_k=seed_k, seed_c=seed_c)
    n = km.numel()
    g = torch.Generator(device=dev)
    g.manual_seed(7 + rank)
    nq = n
    perm = torch.randperm(n, device=dev, generator=g)[:nq]
    q = km[perm].clone()
    q[: nq // 2] = synth_torch.revcomp(q[: nq // 2], a.k)
    q = torch.cat([q, synth_torch.random_kmers(max(nq // 10, 10), a.k, 0xABCDEF0123 + rank * (1 << 32), dev)])
    out = torch.empty(q.numel(), dtype=torch.int32, device=dev)
    del perm

    m = KModel(a.ci, a.cs, a.nh, a.nb)
    stream = torch.cuda.current_stream().cuda_stream
    m.set_stream(stream)

    def insert_step():
        m.build_dev(a.k, km.data_ptr(), cnt.data_ptr(), n)

    def query_step():
        m.kmer_to_occ_dev(q.data_ptr(), q.numel(), out.data_ptr())

    for _ in range(a.warmup):
        insert_step()
        query_step()
    sync_all(distributed)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        insert_step()
    sync_all(distributed)
    t_ins = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        query_step()
    sync_all(distributed)
    t_q = time.perf_counter() - t0
    (t_ins, t_q), (n_all, nq_all) = kd.reduce_job([t_ins, t_q], [n, q.numel()], device="cpu" if rehearsal else dev)   # MAX of times, SUM of units
    st = m.stats()

    # ---- roofline leg: same steps again with HIP events around every launch of each kernel class
    roof = None
    extra = {}
    if rank == 0 and not a.no_roofline:
        m.set_profile(True)
        m.kernel_times(reset=True)
        for _ in range(a.steps):
            insert_step()
        for _ in range(a.steps):
            query_step()
        kt = m.kernel_times(reset=True)
        m.set_profile(2)                                                 # one more build with the fused launches' accounting variant (never timed:
        insert_step()                                                    # its counters cost the kernel 8 %): what they examined, committed, issued
        torch.cuda.synchronize()
        stp = m.stats()
        m.set_profile(False)
        A, S, nbf = st.attempts, st.successes, sum(st.n_bf)
        W8 = 8 * ((a.k + 31) // 32) + 4
        Ap, Sp = stp.piped_attempts, stp.piped_commits                   # the part of A and S handled inside the fused commit|check launches
        alg = {  # algorithmic bytes of ONE step per kernel class (SURVEY.md §8d formula, split by kernel)
            "check": G * (A - Ap) * a.nh + (A - Ap) * W8,                # reads: every attempt looks at its nh positions
            "commit": G * max(st.fast_commits - Sp, 0) * a.nh,           # write-backs in launches of their own (km_back's 2*S*(nh-2) term is k_kmback_emit + k_bs_apply now)
            "commit_check": G * Ap * a.nh + Ap * W8 + G * Sp * a.nh,     # the winners of round r-1 commit beside the check of round r
            "classify": G * 2 * nbf * ((a.nh - 1) + (a.nh - 2)) + n * W8,
        }
        per_q = G * (a.nb * a.nh + (a.nh - 2) + 2 + 6) + W8      # §8d: ~48 touches per query
        alg["query"] = q.numel() * per_q
        total_insert_alg = alg["check"] + alg["commit"] + alg["commit_check"] + G * 2 * S * (a.nh - 2) + alg["classify"]     # the §8d formula, whole insert
        classes = {}
        for name, v in kt.items():
            if v["launches"]:
                classes[name] = {"launches_per_step": v["launches"] / a.steps, "seconds_per_step": v["seconds"] / a.steps,
                                 "avg_launch_us": v["seconds"] / v["launches"] * 1e6}
                if name in alg:
                    classes[name]["alg_GBps"] = alg[name] / (v["seconds"] / a.steps) / 1e9
        dom = max((c for c in classes if c != "query"), key=lambda c: classes[c]["seconds_per_step"])
        dv = kt[dom]
        per_launch = alg.get(dom, 0) * a.steps / max(dv["launches"], 1)
        ach = per_launch / (dv["seconds"] / max(dv["launches"], 1)) / 1e9
        # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # runs of this same command, tools/pmc_summary.py); only valid for the default workload they were taken on
        traffic, traffic_src, traffic_head, q_pmc = None, None, None, None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic_default_workload.json")
        kname = {"check": "k_round_check_emit<", "commit": "k_round_commit<", "commit_check": "k_round_commit_check<", "classify": "k_classify_count"}.get(dom)
        if a.n == 100_000_000 and (a.k, a.nh, a.nb, a.ci, a.cs) == (31, 7, 5, 1, 1023) and os.path.exists(pmc_file) and kname:
            pm = json.load(open(pmc_file))
            traffic_head = pm.get("head")                          # the commit the counters were taken at
            if pm.get("src_sha") != source_digest():               # counters of other code say nothing about this run: traffic stays null
                traffic_src = f"stale: {os.path.basename(pmc_file)} was taken on sources {pm.get('src_sha')}, this run is {source_digest()}"
                pm = {"kernels": {}}
            for kn, kv in pm["kernels"].items():
                if kn.startswith(kname):
                    traffic, traffic_src = kv["hbm_bytes_per_launch"], "profiles/pmc_traffic_default_workload.json"
                if kn.startswith("k_query<"):                      # probes counted by the memory system: fetched bytes / calibrated bytes per random load
                    per_load = pm.get("calibration", {}).get("bytes_fetched_per_random_4B_load") or pm.get("calibration", {}).get("bytes_fetched_per_random_8B_load")
                    if per_load and pm.get("n_queries"):
                        q_pmc = kv["fetch_bytes_total"] / max(kv["launches"], 1) / per_load / pm["n_queries"]
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                "traffic": traffic, "traffic_source": traffic_src, "traffic_head": traffic_head, "alg_bytes_per_launch": per_launch,
                "avg_launch_us": classes[dom]["avg_launch_us"], "launches": dv["launches"]}
        if dom == "commit_check":
            # what the fused launches actually ISSUED (device counters): the staged fetch stops at the first conflicting group and a
            # winner sets only the positions it saw untagged -- priced like the formula at G bytes per random operation
            issued = (stp.piped_gathers + stp.piped_atomics) * G + Ap * W8
            roof["issued_frac"] = issued / (dv["seconds"] / a.steps) / 8e12
            roof["issued"] = {"gathers_per_step": stp.piped_gathers, "atomics_per_step": stp.piped_atomics,
                              "random_ops_per_s": (stp.piped_gathers + 2 * stp.piped_atomics) / (dv["seconds"] / a.steps),
                              "note": "random_ops_per_s counts an atomic as a read and a write at the DRAM (DESIGN.md 4: one budget of ~56 G/s)"}
        # the random-access ceiling of this chip for 8-byte touches over a footprint like the coupled arrays'
        foot = max(int(st.km_byte_size) * 2 * a.nb, 1 << 26)              # the cells of all arrays (4 bytes per 16 positions)
        tg = api.microbench(8, foot, 1 << 27, 3)                           # 4-byte random loads: what check_emit issues
        ta = api.microbench(5, foot, 1 << 27, 3)                           # 32-bit random atomic ORs: what commit issues
        tp = api.microbench(20, foot, 1 << 27, 3)                          # both side by side on two streams: ONE budget of random DRAM operations
        pair_ops = 3.0 * (1 << 27) / tp                                    # a gather = 1 operation, an atomic = a read + a write
        if traffic:
            roof["traffic_frac"] = traffic / (dv["seconds"] / max(dv["launches"], 1)) / 8e12     # what the memory system really moved per launch / its time / 8 TB/s
        roof["note"] = ("achieved prices a touched position at G = 32 B (SURVEY 8d); at the fabric a random 4-byte load fetches 64 B "
                        "(calibration in the PMC file), so traffic > alg_bytes_per_launch is the access granule, not re-reads: traffic_frac is the HBM rate")
        if dom == "commit_check" and stp.piped_gathers:
            ops = stp.piped_gathers + 2 * stp.piped_atomics                # the random DRAM operations a whole build needs (its fused launches issue all of them)
            roof["budget_frac_whole_build"] = ops / (t_ins / a.steps) / pair_ops
            roof["budget"] = {"random_ops_per_build": ops, "pair_ceiling_ops_per_s": pair_ops, "build_ops_per_s": ops / (t_ins / a.steps),
                              "fused_launches_ops_per_s": ops / (dv["seconds"] / a.steps),
                              "note": "gathers + 2 x atomics of one build / ms_per_step, against gathers and atomics run side by side in this same run (kmx_microbench mode 20)"}
        roof["random_access_ceiling_GBps_at_32B"] = (1 << 27) * G / tg / 1e9     # measured 4-byte gather rate, priced at G per touch
        roof["frac_of_random_access_ceiling"] = ach / roof["random_access_ceiling_GBps_at_32B"]
        extra = {"kernel_classes": classes,
                 "insert_alg_bytes_per_kmer": total_insert_alg / n,
                 "insert_alg_GBps_whole_step": total_insert_alg / (t_ins / a.steps) / 1e9,
                 "insert_frac_of_8TBps": total_insert_alg / (t_ins / a.steps) / 8e12,
                 "query_alg_GBps": alg["query"] / (t_q / a.steps) / 1e9,
                 "query_touches_per_query_pmc": q_pmc,                                    # what the counters saw (early exits)
                 "query_frac_of_8TBps_counted": (q_pmc * G * q.numel() / (t_q / a.steps) / 8e12) if q_pmc else None,
                 "random_access_ceiling": {"footprint_bytes": foot, "gather_Gtouch_s": (1 << 27) / tg / 1e9,
                                           "atomic_or_Gtouch_s": (1 << 27) / ta / 1e9,
                                           "gather_GBps_at_32B": (1 << 27) * G / tg / 1e9,
                                           "atomic_GBps_at_32B": (1 << 27) * G / ta / 1e9}}
    init_db = {}
    if rank == 0 and world == 1 and not a.no_init_db and a.k <= 31:
        try:
            init_db = init_db_leg(a, km, cnt)
        except Exception as e:  # noqa: BLE001
            init_db = {"init_db_error": repr(e)}
    if rank == 0 and world == 1 and not a.no_init_db and a.k <= 31:
        try:
            init_db.update(init_db_leg(a, km, cnt, layout="kmc2"))
            init_db["init_db_kmc2_over_kmc1"] = init_db["init_db_kmc2_value"] / init_db["init_db_value"] if init_db.get("init_db_value") else None
        except Exception as e:  # noqa: BLE001
            init_db["init_db_kmc2_error"] = repr(e)
    if rank == 0 and world == 1 and a.genome_bases > 0 and a.k <= 31:
        try:
            init_db.update(genome_leg(a, dev, a.genome_bases))
            init_db["query_genome_over_random"] = init_db["query_genome_value"] / (nq_all * a.steps / t_q)
        except Exception as e:  # noqa: BLE001
            init_db["genome_error"] = repr(e)
    if rank == 0 and world == 1 and not a.no_query_strings and a.k <= 31:
        try:
            query_step()
            torch.cuda.synchronize()
            init_db.update(query_strings_leg(a, m, q, out))
        except Exception as e:  # noqa: BLE001
            init_db["query_strings_error"] = repr(e)
    # ---- ONE model over all ranks' streams; a watchdog ends the job with what is measured if the exchange hangs
    single = None
    if not a.no_single_model:
        import threading
        done = threading.Event()

        def give_up():
            if done.is_set():
                return
            if rank == 0:
                partial["single_model"] = {"error": "no result within the watchdog limit: exchange presumed hung"}
                if world > 1:
                    partial["replica_value"], partial["value"] = partial["value"], None     # the headline (one model) was NOT measured
                emit_line(partial)
            os._exit(3)                                                 # a hung exchange is a failed run on every rank
        partial = {}
        if rank == 0:
            partial = headline(a, world, n, q, n_all, nq_all, t_ins, t_q, st, roof, None, extra, init_db)
        wd = threading.Timer(400.0 + 40.0 * (a.single_model_steps + a.warmup), give_up)
        wd.daemon = True
        wd.start()
        try:
            single = single_model_leg(a, m, km, cnt, q, out, rank, world, dev, rehearsal, distributed)
        except Exception as e:  # noqa: BLE001
            single = {"error": repr(e)}
        # N > 1: the OTHER partition too, a few builds, beside the headline (not in it): the first run on a node times both
        if world > 1 and "value" in single and not a.no_other_partition:
            import copy
            b = copy.copy(a)
            b.partition = "range" if a.partition == "ring" else "ring"
            b.single_model_steps, b.warmup = min(a.single_model_steps, 3), 1
            try:
                o = single_model_leg(b, m, km, cnt, q, out, rank, world, dev, rehearsal, distributed)
                single["other_partition"] = {k: o[k] for k in o if k not in ("what", "stats")}
            except Exception as e:  # noqa: BLE001
                single["other_partition"] = {"partition": b.partition, "error": repr(e)}
        # N > 1: the C++ entry too -- ONE process (rank 0's) drives all N GPUs with kmx_build_from_kmc_multi_ex while the other ranks
        # wait: KModel::init on rank 0's stream as a KMC database (strong scaling: the same 1e8 k-mers whatever N is), through the
        # peer-mapped inboxes, through RCCL messages, and as the ring -- the transports this pool's one GPU cannot time
        # (rehearsal on one GPU: only with KMX_BENCH_CXX_MULTI=1, every handle on cuda:0 -- the code path, not a rate)
        # It runs in a CHILD of rank 0 (a fresh process: its own HIP context, its own RCCL communicators, a time limit): whatever
        # happens there -- a refused peer mapping, a hung collective -- costs this leg, not the run.
        if world > 1 and (not rehearsal or os.environ.get("KMX_BENCH_CXX_MULTI") == "1") and not a.no_cxx_multi and single is not None and a.k <= 31:
            if rank == 0:
                devs = [0] * world if rehearsal else list(range(world))
                cmd = [sys.executable, os.path.abspath(__file__), "--cxx-multi-child", ",".join(str(d) for d in devs), "--kmers", str(a.n), "--kmer-len", str(a.k),
                       "--ci", str(a.ci), "--cs", str(a.cs), "--nh", str(a.nh), "--nb", str(a.nb)]
                try:
                    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                    single["cxx_multi"] = json.loads(lines[-1]) if p.returncode == 0 and lines else {"error": f"child exit {p.returncode}: {p.stderr[-400:]}"}
                except subprocess.TimeoutExpired:
                    single["cxx_multi"] = {"error": "no result within 300 s: the child was ended"}
                except Exception as e:  # noqa: BLE001
                    single["cxx_multi"] = {"error": repr(e)}
            dist.barrier()
        done.set()
        wd.cancel()
    cpu = None
    if rank == 0 and world == 1 and a.cpu_sample > 0:
        try:
            cpu = cpu_baseline(a)
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)}
    failed = world > 1 and single is not None and "value" not in single
    if rank == 0:
        line = headline(a, world, n, q, n_all, nq_all, t_ins, t_q, st, roof, cpu, extra, init_db)
        emit_line(promote_single_model(line, single, world))
    if distributed:
        dist.barrier()                      # rank 0 may still have been in its roofline leg: leave together
        dist.destroy_process_group()
    if failed:
        sys.exit(3)                         # like the watchdog: a failed exchange is a failed run, on every rank


if __name__ == "__main__":
    main()
