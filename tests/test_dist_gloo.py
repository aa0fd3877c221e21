"""CPU, world_size 2 over gloo: the N>1 plumbing of bench.py (per-rank seeds, batch split, timing/count reduction,
ragged answer gather).  The data path itself has no collective (one model per rank)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kmcex_amd import dist as kd
from kmcex_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert kd.env_world() == (rank, rank, world)
        sk, sc = kd.stream_seeds(rank)
        km, cnt = synth.make_stream(2000, 31, 1, 1023, seed_k=sk, seed_c=sc)
        t, u = kd.reduce_job([0.5 + rank, 2.0 - rank], [len(cnt), 7])
        # a batch of 11 answers split over 2 ranks, gathered back in order
        n = 11
        lo, hi = kd.split_batch(n, world, rank)
        local = torch.arange(lo, hi, dtype=torch.int32) * 10
        full = kd.gather_slices(local, n, world, rank)
        q.put((rank, t, u, full.tolist(), int(km[0]), len(cnt)))
    finally:
        dist.destroy_process_group()


def test_two_rank_reduction_and_split():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, t0, u0, f0, k0, n0), (r1, t1, u1, f1, k1, n1) = res
    assert t0 == t1 == [1.5, 2.0]                      # MAX over ranks
    assert u0 == u1 == [n0 + n1, 14]                   # SUM over ranks
    assert f0 == f1 == [10 * i for i in range(11)]     # ragged gather restores batch order
    assert k0 != k1                                    # ranks draw different streams


def test_split_batch_covers_everything_once():
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [kd.split_batch(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert kd.stream_seeds(0) == (1, 2)
