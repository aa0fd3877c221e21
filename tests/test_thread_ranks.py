"""The UNSTAGED path of kmcex_amd.dist (what "nccl" runs: tensors straight to the transport, whole ring messages through
`Comm.exchange`) with more than one rank, on the CPU: ranks are threads of this process (tests/thread_comm.py), the per-rank
engine is the oracle.  The gloo tests cover the staged path (`exchange_counted`); the GPU twin of this test is
test_gpu_dist.py::test_many_ranks_in_one_process_are_bit_exact."""
import os

import pytest
import torch

from common import sha_file
from dist_workers import _to_torch, listing_of
from kmcex_amd import dist as kd
from oracle_engine import OracleEngine
from range_engine import RangeOracleEngine
from thread_comm import run_threads


@pytest.mark.parametrize("partition,world", [("ring", 8), ("ring", 3), ("range", 8), ("range", 16)])
def test_unstaged_orchestration_over_thread_ranks(partition, world, golden, tmp_path):
    torch.set_num_threads(1)
    g = golden["cases"]["tiny_k31"]
    k, ci, cs, nh, nb, km, cnt, _ = listing_of(("synth", "tiny_k31"))

    def body(rank, comm):
        lo, hi = kd.split_batch(len(cnt), world, rank)
        tk, tc = _to_torch(km[lo:hi], cnt[lo:hi], k, "cpu")
        eng = (RangeOracleEngine if partition == "range" else OracleEngine)(ci, cs, nh, nb)
        info = kd.build_sharded(eng, comm, k, nb, eng.bf_num, tk, tc, partition=partition)
        d = os.path.join(str(tmp_path), f"rank{rank}")
        os.makedirs(d)
        eng.o.save(d)
        return info, {f: sha_file(os.path.join(d, f)) for f in ("header", "km.bin", "rest.bin")}

    for rank, (info, sha) in enumerate(run_threads(world, body)):
        assert sha == {f: g["sha256"][f] for f in ("header", "km.bin", "rest.bin")}, f"rank {rank} holds a different model"
