"""ONE model built and queried by several ranks: kmcex_amd.dist over libkmx.so, the ranks sharing cuda:0
and exchanging over gloo (the rehearsal transport: RCCL refuses two ranks on one device; on a multi-GPU node the same
orchestration runs over "nccl").  The bar is the single-GPU bar: the files EVERY rank saves are the reference's files,
the statistics are the sequential ones, the replica query gives the golden answers."""
import os

import numpy as np
import pytest

from dist_workers import gpu_worker, rccl_worker, run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden_of(golden, spec):
    sect = {"synth": "cases", "kmc2": "kmc2_cases", "genome": "genome_cases"}[spec[0]]
    return golden[sect][spec[1]]


# world sizes: 2 (uneven runs of arrays), nb (one array per rank) or the most ranks the box lets share a GPU (5 + pytest)
@pytest.mark.parametrize("spec,world", [
    (("synth", "tiny_k31"), 2),
    (("synth", "k31_multiblock_ci1"), 2),          # 2 full blocks + a partial block with unused rows: quirk Q1 across ranks
    (("synth", "k31_multiblock_ci1"), 5),          # nb ranks: the reference's rotation, one array per GPU
    (("synth", "k55_multiblock"), 2),              # two-word k-mers, nh 9, nb 6
    (("synth", "k55_multiblock"), 5),              # 6 arrays on 5 ranks
    (("kmc2", "k31_kmc2_6bins"), 2),               # bin-major (unsorted) listing order
    (("kmc2", "k31_kmc2_6bins"), 3),
    (("synth", "k31_multiblock_ci2"), 4),          # three Bloom classes merged by OR
    (("synth", "k27_ci3_nb8"), 3),                 # 8 arrays on 3 ranks, a single partial block
    (("synth", "k31_nh3_nb1"), 2),                 # one array: rank 1 only lists and classifies
    (("genome", "genome_k31_ci1"), 3),
], ids=lambda v: v[1] if isinstance(v, tuple) else f"w{v}")
def test_single_model_over_ranks_is_bit_exact(spec, world, golden, tmp_path):
    g = _golden_of(golden, spec)
    res = run_ranks(gpu_worker, world, spec, str(tmp_path))
    for r in res:
        assert r["sha"] == {f: g["sha256"][f] for f in ("header", "km.bin", "rest.bin")}, f"rank {r['rank']} holds a different model"
        assert r["occ_sha"] == g["occ_sha256"]
        if "stats" in g:
            assert r["stats"][2:5] == (g["stats"]["attempts"], g["stats"]["successes"], g["stats"]["rest_entries"])
    owned = sorted(a for r in res for a in r["info"]["arrays_owned"])
    assert owned == list(range(len(owned)))                        # every array has exactly one owner


# The north star's partition: every array cut by position range, triples / verdicts / commits by all-to-all --
# as fixed-size messages with the counts in band (no host wait in a round; the default), as counted ones, and with regions forced
# far too small (KMX_RANGE_CAPX under KMX_TEST_HOOKS): words are dropped, every rank learns it, the build is repeated, counted.
@pytest.mark.parametrize("spec,world,messages", [
    (("synth", "tiny_k31"), 1, "fixed"),           # one rank: the kernels of the partition alone, every word "sent" to itself
    (("synth", "tiny_k31"), 2, "fixed"),
    (("synth", "k31_multiblock_ci1"), 2, "fixed"), # 3 lists on rank 0, 2 on rank 1; quirk Q1 on the ranks that hold the unused rows
    (("synth", "k31_multiblock_ci1"), 3, "fixed"),
    (("synth", "k31_multiblock_ci1"), 5, "fixed"),
    (("synth", "k55_multiblock"), 2, "fixed"),     # two-word k-mers, nh 9 (16-wide templates), nb 6
    (("synth", "k55_multiblock"), 5, "fixed"),
    (("kmc2", "k31_kmc2_6bins"), 3, "fixed"),      # bin-major (unsorted) listing order
    (("synth", "k31_multiblock_ci2"), 2, "fixed"), # three Bloom classes
    (("synth", "k31_nh3_nb1"), 3, "fixed"),        # one list, three range owners
    (("synth", "k31_multiblock_ci1"), 3, "counted"),
    (("synth", "k55_multiblock"), 2, "counted"),
    (("synth", "k31_multiblock_ci1"), 2, "overflow"),
    (("synth", "k31_multiblock_ci2"), 3, "overflow"),
], ids=lambda v: v[1] if isinstance(v, tuple) else str(v))
def test_range_partition_over_ranks_is_bit_exact(spec, world, messages, golden, tmp_path, monkeypatch):
    g = _golden_of(golden, spec)
    if messages == "counted":
        monkeypatch.setenv("KMX_RANGE_MESSAGES", "counted")
    if messages == "overflow":
        monkeypatch.setenv("KMX_TEST_HOOKS", "1")
        monkeypatch.setenv("KMX_RANGE_CAPX", "30000")
    res = run_ranks(gpu_worker, world, spec, str(tmp_path), None, "range")
    for r in res:
        assert r["sha"] == {f: g["sha256"][f] for f in ("header", "km.bin", "rest.bin")}, f"rank {r['rank']} holds a different model"
        assert r["occ_sha"] == g["occ_sha256"]
        if "stats" in g:
            assert r["stats"][2:5] == (g["stats"]["attempts"], g["stats"]["successes"], g["stats"]["rest_entries"])
        assert r["info"]["partition"] == "range" and r["info"]["collectives"] >= (2 * r["info"]["blocks"] * (len(res) > 1))
        assert r["info"]["messages"] == ("fixed" if messages == "fixed" else "counted")
        assert bool(r["info"].get("fixed_messages_overflowed")) == (messages == "overflow")
    cells = sorted(tuple(r["info"]["cells_owned"]) for r in res)
    assert cells[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cells, cells[1:]))       # the ranges tile every array


# The world size the driver's 8-GPU node starts with (and the 16 the range partition is sized for): more ranks than the box
# lets share the card as processes, so here they are THREADS of one process (tests/thread_comm.py) on the UNSTAGED path of the
# orchestration -- device tensors straight to the transport, whole ring messages -- which is the code "nccl" runs.
@pytest.mark.parametrize("spec,world,partition", [
    (("synth", "k31_multiblock_ci1"), 8, "ring"),  # 5 array owners, 3 ranks that only list, classify and merge
    (("synth", "k31_multiblock_ci1"), 8, "range"), # 5 list holders, 8 range owners
    (("synth", "k55_multiblock"), 8, "range"),     # two-word k-mers, nh 9, nb 6
    (("synth", "k55_multiblock"), 8, "ring"),
    (("synth", "k31_multiblock_ci2"), 16, "range"),  # KMX_MAX_RANKS owners, three Bloom classes
    (("kmc2", "k31_kmc2_6bins"), 7, "range"),      # a world that divides nothing
], ids=lambda v: v[1] if isinstance(v, tuple) else str(v))
def test_many_ranks_in_one_process_are_bit_exact(spec, world, partition, golden, tmp_path):
    import torch
    from common import sha_file, sha_occ
    from dist_workers import _to_torch, listing_of, queries_of
    from kmcex_amd import KModel
    from kmcex_amd import dist as kd
    from thread_comm import run_threads
    g = _golden_of(golden, spec)
    dev = torch.device("cuda", 0)
    k, ci, cs, nh, nb, km, cnt, base = listing_of(spec)
    qk = queries_of(spec, base, k)
    tq = torch.from_numpy(np.ascontiguousarray(qk, dtype=np.uint64).view(np.int64).reshape(-1)).to(dev)

    def rank_body(rank, comm):
        lo, hi = kd.split_batch(len(cnt), world, rank)
        tk, tc = _to_torch(km[lo:hi], cnt[lo:hi], k, dev)
        m = KModel(ci, cs, nh, nb)
        try:
            info = kd.build_sharded(kd.DeviceEngine(m, dev), comm, k, nb, 1 if ci == 1 else 3, tk, tc, partition=partition)
            st = m.stats()
            occ = kd.query_replicas(m, comm, tq, k).cpu().numpy()
            d = os.path.join(str(tmp_path), f"rank{rank}")
            os.makedirs(d, exist_ok=True)
            m.save(d)
            return {"info": info, "stats": (st.attempts, st.successes, st.rest_entries), "occ_sha": sha_occ(occ),
                    "sha": {f: sha_file(os.path.join(d, f)) for f in ("header", "km.bin", "rest.bin")}}
        finally:
            m.close()

    res = run_threads(world, rank_body)
    for rank, r in enumerate(res):
        assert r["sha"] == {f: g["sha256"][f] for f in ("header", "km.bin", "rest.bin")}, f"rank {rank} holds a different model"
        assert r["occ_sha"] == g["occ_sha256"]
        if "stats" in g:
            assert r["stats"] == (g["stats"]["attempts"], g["stats"]["successes"], g["stats"]["rest_entries"])
    if partition == "ring":
        owned = sorted(a for r in res for a in r["info"]["arrays_owned"])
        assert owned == list(range(nb))
    else:
        cells = sorted(tuple(r["info"]["cells_owned"]) for r in res)
        assert cells[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cells, cells[1:]))


def test_replica_query_of_reference_files():
    """2 ranks load the model the REFERENCE wrote (tests/golden/tiny) and answer the batch by slices."""
    d = os.path.join(ROOT, "tests", "golden", "tiny")
    exp = np.loadtxt(os.path.join(d, "occ.txt"), dtype=np.int32).tolist()
    res = run_ranks(gpu_worker, 2, None, None, d)
    for r in res:
        assert r["occ"] == exp


def test_collectives_through_rccl():
    """every torch.distributed call of the sharded build, on the "nccl" (RCCL) backend with the single rank this box has"""
    res = run_ranks(rccl_worker, 1, None, None, timeout=300)
    assert res[0]["ok"]
