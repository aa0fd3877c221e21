"""Worker processes of the multi-rank tests (spawned, one per rank): a GPU worker that runs the product
(kmcex_amd.dist over libkmx.so, several ranks sharing cuda:0, exchange over gloo = the rehearsal transport) and a CPU
worker that drives the SAME orchestration with an engine made of the oracle (tests/oracle_engine.py)."""
import os
import socket
import sys
import tempfile
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def listing_of(spec):
    """(k, ci, cs, nh, nb, kmers, counts, kmers the golden query set is drawn from); kmers / counts in LISTING order; spec =
    ("synth", name) | ("kmc2", name) | ("genome", name) | ("raw", k, ci, cs, nh, nb, n, seed_k, seed_c, min_count)"""
    from common import CASE, GENOME_CASES, KMC2_CASES
    from kmcex_amd import api, kmcdb, synth
    kind = spec[0]
    if kind == "synth":
        _, k, ci, cs, nh, nb, n = CASE[spec[1]]
        km, cnt = synth.make_stream(n, k, ci, cs)
        base = km
    elif kind == "kmc2":
        name, k, ci, cs, nh, nb, n, n_bins = [c for c in KMC2_CASES if c[0] == spec[1]][0]
        km, cnt = synth.make_stream(n, k, ci, cs)
        base = km
        with tempfile.TemporaryDirectory(prefix="kmx_dist_") as tmp:           # the database lists bin-major: not sorted
            kmcdb.write_kmc2(os.path.join(tmp, "db"), km, cnt, k, ci, cs, n_bins=n_bins)
            _, _, km, cnt = api.kmc_list(os.path.join(tmp, "db"))
    elif kind == "genome":
        name, k, ci, cs, nh, nb, n_bases = [c for c in GENOME_CASES if c[0] == spec[1]][0]
        km, cnt = synth.genome_stream(n_bases, k, ci, cs)
        base = km
    else:
        _, k, ci, cs, nh, nb, n, seed_k, seed_c, min_count = spec
        km, cnt = synth.make_stream(n, k, ci, cs, seed_k=seed_k, seed_c=seed_c)
        cnt = np.maximum(cnt, min_count).astype(np.uint32)
        base = km
    return k, ci, cs, nh, nb, km, cnt, base


def queries_of(spec, km, k):
    from common import genome_query_set, query_set
    return genome_query_set(km, k) if spec[0] == "genome" else query_set(km, k)


def _to_torch(km, cnt, k, device):
    import torch
    W = (k + 31) // 32
    tk = torch.from_numpy(np.ascontiguousarray(km, dtype=np.uint64).view(np.int64).reshape(-1, W) if W > 1 else np.ascontiguousarray(km, dtype=np.uint64).view(np.int64))
    tc = torch.from_numpy(np.ascontiguousarray(cnt, dtype=np.uint32).view(np.int32))
    return tk.to(device), tc.to(device)


def gpu_worker(rank, world, port, spec, out_dir, q, load_dir=None, partition="ring"):
    """One rank of a single-model build (or, with load_dir, of a replica query) sharing cuda:0 with the others."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import torch
        import torch.distributed as dist
        from common import sha_file, sha_occ
        from kmcex_amd import KModel
        from kmcex_amd import dist as kd
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = kd.Comm()
        res = {"rank": rank}
        if load_dir is not None:                                    # replica query of a model directory (every rank loads it)
            from kmcex_amd import synth
            m = KModel.load(load_dir)
            qs = open(os.path.join(load_dir, "queries.txt")).read().split()
            qk = synth.from_strings(qs, 31).reshape(-1)
            tq = torch.from_numpy(qk.view(np.int64)).to(dev)
            occ = kd.query_replicas(m, comm, tq, 31).cpu().numpy()
            res["occ"] = occ.tolist()
        else:
            k, ci, cs, nh, nb, km, cnt, base = listing_of(spec)
            lo, hi = kd.split_batch(len(cnt), world, rank)          # this rank lists a contiguous slice of the database
            tk, tc = _to_torch(km[lo:hi], cnt[lo:hi], k, dev)
            m = KModel(ci, cs, nh, nb)
            eng = kd.DeviceEngine(m, dev)
            info = kd.build_sharded(eng, comm, k, nb, 1 if ci == 1 else 3, tk, tc, partition=partition)
            st = m.stats()
            res.update(info=info, stats=(st.n_km, list(st.n_bf), st.attempts, st.successes, st.rest_entries, st.blocks, st.rounds))
            qk = queries_of(spec, base, k)
            tq = torch.from_numpy(np.ascontiguousarray(qk, dtype=np.uint64).view(np.int64).reshape(-1)).to(dev)
            occ = kd.query_replicas(m, comm, tq, k).cpu().numpy()
            res["occ_sha"] = sha_occ(occ)
            d = os.path.join(out_dir, f"rank{rank}")                # EVERY rank saves: each must hold the whole model
            os.makedirs(d, exist_ok=True)
            m.save(d)
            res["sha"] = {f: sha_file(os.path.join(d, f)) for f in ("header", "km.bin", "rest.bin")}
        m.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put(res)
    except Exception:  # noqa: BLE001
        q.put({"rank": rank, "error": traceback.format_exc()})


def cpu_worker(rank, world, port, spec, out_dir, q, partition="ring"):
    """One rank of the same orchestration (kmcex_amd.dist.build_sharded) over gloo with the ORACLE as the per-rank engine:
    checks the protocol -- routing all-to-all, ring of arrays, OR-merge, survivor gather -- without a GPU."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import torch
        import torch.distributed as dist
        torch.set_num_threads(1)
        os.environ["OMP_NUM_THREADS"] = "1"
        from common import sha_file, sha_occ
        from kmcex_amd import dist as kd
        from oracle_engine import OracleEngine
        from range_engine import RangeOracleEngine
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = kd.Comm()
        k, ci, cs, nh, nb, km, cnt, base = listing_of(spec)
        lo, hi = kd.split_batch(len(cnt), world, rank)
        tk, tc = _to_torch(km[lo:hi], cnt[lo:hi], k, "cpu")
        eng = (RangeOracleEngine if partition == "range" else OracleEngine)(ci, cs, nh, nb)
        info = kd.build_sharded(eng, comm, k, nb, eng.bf_num, tk, tc, partition=partition)
        so = eng.o.stats()
        res = {"rank": rank, "info": info, "stats": (so.n_km, list(so.n_bf), so.attempts, so.successes, so.rest_entries)}
        d = os.path.join(out_dir, f"rank{rank}")
        eng.o.save(d)
        res["sha"] = {f: sha_file(os.path.join(d, f)) for f in ("header", "km.bin", "rest.bin")}
        if rank == 0:
            res["occ_sha"] = sha_occ(eng.o.query_packed(k, queries_of(spec, base, k), threads=2))
        dist.barrier()
        dist.destroy_process_group()
        q.put(res)
    except Exception:  # noqa: BLE001
        q.put({"rank": rank, "error": traceback.format_exc()})


def rccl_worker(rank, world, port, spec, out_dir, q):
    """The collectives of kmcex_amd.dist.Comm through the REAL backend of a multi-GPU node ("nccl" = RCCL) with the one
    rank a one-GPU box allows: dtypes, ragged splits and in-place semantics of every call the sharded build makes."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import torch
        import torch.distributed as dist
        from kmcex_amd import KModel
        from kmcex_amd import dist as kd
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        comm = kd.Comm(shortcut=False)
        assert not comm.staged
        g = torch.Generator(device="cpu").manual_seed(3)
        km = torch.randint(-2**62, 2**62, (1000, 2), dtype=torch.int64, generator=g).to(dev)
        cn = torch.randint(0, 1000, (1000,), dtype=torch.int32, generator=g).to(dev)
        assert comm.all_reduce_ints([3, 4, 5], dev) == [3, 4, 5]
        assert comm.all_gather_ints(7, dev) == [7]
        assert torch.equal(comm.all_to_all_v(km, [1000], [1000]), km)
        assert torch.equal(comm.all_to_all_v(cn[:0], [0], [0]), cn[:0])
        assert torch.equal(comm.all_gather_v(km[:17], [17]), km[:17])
        assert torch.equal(comm.all_gather_v(cn[:0], [0]), cn[:0])
        r = cn.clone()
        comm.all_gather_ranges(r, [0, 1000], window=300)             # the range partition's array replication: in place, in windows (4 collectives here)
        assert torch.equal(r, cn)
        b = cn.clone()
        comm.broadcast(b, 0)
        assert torch.equal(b, cn)
        # the ring's hand-off (batch_isend_irecv): with one rank, a message sent to itself through RCCL
        got = torch.zeros_like(km)
        comm.exchange([(km, 0)], [(got, 0)])
        assert torch.equal(got, km)
        m = KModel(1, 1023, 7, 5)
        eng = kd.DeviceEngine(m, dev)
        w = cn[:999].clone()                                        # odd length: exercises the padding of the range split
        comm.or_allreduce(w, eng.or_into)
        assert torch.equal(w, cn[:999])
        m.close()
        # the range partition's exchanges: split sizes, 64-bit words, verdict bytes -- and one whole build through them
        assert comm.all_to_all_ints([5], dev) == [5]
        by = (cn[:300] & 7).to(torch.uint8)
        assert torch.equal(comm.all_to_all_v(by, [300], [300]), by)
        from common import CASE, sha_file
        from kmcex_amd import synth
        import json, tempfile
        _, k, ci, cs, nh, nb, n = CASE["k31_multiblock_ci1"]
        skm, scnt = synth.make_stream(n, k, ci, cs)
        tk, tc = _to_torch(skm, scnt, k, dev)
        m = KModel(ci, cs, nh, nb)
        info = kd.build_sharded(kd.DeviceEngine(m, dev), comm, k, nb, 1, tk, tc, partition="range")
        assert info["collectives"] >= 2 * info["blocks"] * nb
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["cases"]["k31_multiblock_ci1"]
        with tempfile.TemporaryDirectory(prefix="kmx_rccl_") as d:
            m.save(d)
            assert {f: sha_file(os.path.join(d, f)) for f in ("header", "km.bin", "rest.bin")} == {f: g["sha256"][f] for f in ("header", "km.bin", "rest.bin")}
        m.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put({"rank": rank, "ok": True})
    except Exception:  # noqa: BLE001
        q.put({"rank": rank, "error": traceback.format_exc()})


def run_ranks(target, world, *args, timeout=600):
    """spawn `world` ranks of `target(rank, world, port, *args, q)`; returns their result dicts in rank order"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=target, args=(r, world, port) + tuple(args[:2]) + (q,) + tuple(args[2:])) for r in range(world)]
    for p in ps:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=timeout))
            if "error" in res[-1]:                                  # the others would wait for it in a collective
                break
    finally:
        failed = any("error" in r for r in res) or len(res) < world
        for p in ps:
            p.join(timeout=1 if failed else 60)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    res.sort(key=lambda r: r["rank"])
    for r in res:
        assert "error" not in r, r["error"]
    return res
