"""`kmcex_amd.dist.Comm` for ranks that are THREADS of one process: every collective is a deposit into a shared table between
two barriers.  It lets one process on the one GPU of the pool run the sharded build with 8 (or 16) ranks -- more than the
box lets share the card as processes, and the world size the driver's 8-GPU node starts with -- over the UNSTAGED code
path of the orchestration (device tensors handed to the transport as they are, whole ring messages: what `"nccl"` runs).
All ranks enqueue on the process's default stream, so "rank q's kernel before rank r's read" is the barrier's order."""
import threading

import numpy as np
import torch


class Hub:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slot = [None] * world


class ThreadComm:
    staged = False                                              # device tensors travel as they are (the RCCL path of dist.py)
    on = True
    shortcut = False

    def __init__(self, hub: Hub, rank: int):
        self.hub, self.rank, self.world = hub, rank, hub.world
        self.bytes_sent = 0
        self.collectives = 0

    def _swap(self, mine):
        """deposit -> everybody's deposits (valid until the next _done)"""
        self.hub.slot[self.rank] = mine
        self.hub.barrier.wait()
        return list(self.hub.slot)

    def _done(self):
        self.hub.barrier.wait()

    def all_reduce_ints(self, values, device, op=None):
        import torch.distributed as dist
        allv = self._swap([int(v) for v in values])
        red = max if op == dist.ReduceOp.MAX else sum
        out = [red(col) for col in zip(*allv)]
        self._done()
        return out

    def all_gather_ints(self, value, device):
        out = [int(v) for v in self._swap(int(value))]
        self._done()
        return out

    def all_to_all_ints(self, values, device):
        allv = self._swap([int(v) for v in values])
        out = [allv[q][self.rank] for q in range(self.world)]
        self._done()
        return out

    def all_to_all_v(self, send, send_splits, recv_splits):
        allv = self._swap((send, [int(x) for x in send_splits]))
        parts = []
        for q in range(self.world):
            t, sp = allv[q]
            lo = int(sum(sp[: self.rank]))
            parts.append(t[lo: lo + sp[self.rank]])
            assert sp[self.rank] == int(recv_splits[q]), "split sizes disagree"
        out = torch.cat(parts)                                   # (a copy: the senders reuse their regions)
        if out.is_cuda:
            torch.cuda.synchronize()                             # the copies read the senders' buffers: done before they move on
        self._done()
        self.collectives += 1
        self.bytes_sent += (int(sum(send_splits)) - int(send_splits[self.rank])) * send.element_size() * int(np.prod(send.shape[1:], dtype=np.int64))
        return out

    def all_to_all_fixed(self, send):
        allv = self._swap(send)
        out = torch.stack([allv[q][self.rank] for q in range(self.world)])       # (a copy: the senders reuse their regions)
        if out.is_cuda:
            torch.cuda.synchronize()
        self._done()
        self.collectives += 1
        self.bytes_sent += send.numel() * send.element_size() * (self.world - 1) // self.world
        return out

    def exchange(self, sends, recvs):
        allv = self._swap([(t, d) for t, d in sends])
        taken = {}
        for dst, src in recvs:
            k = taken.get(src, 0)
            msgs = [t for t, d in allv[src] if d == self.rank]
            dst.copy_(msgs[k])
            taken[src] = k + 1
        if any(t.is_cuda for t, _ in recvs):
            torch.cuda.synchronize()
        self._done()
        self.bytes_sent += sum(t.numel() * t.element_size() for t, _ in sends)

    def exchange_counted(self, sends, recvs, pieces):
        raise AssertionError("the unstaged transport ships whole ring messages")

    def broadcast(self, t, src):
        allv = self._swap(t if self.rank == src else None)
        if self.rank != src:
            t.copy_(allv[src])
            if t.is_cuda:
                torch.cuda.synchronize()
        self._done()

    def all_gather_v(self, local, counts):
        allv = self._swap(local)
        out = torch.cat([allv[q][: int(counts[q])] for q in range(self.world)])
        if out.is_cuda:
            torch.cuda.synchronize()
        self._done()
        return out

    def all_gather_ranges(self, v, lo, window=1 << 24):
        allv = self._swap(v)
        for q in range(self.world):
            a, b = int(lo[q]), int(lo[q + 1])
            if q != self.rank and b > a:
                v[a:b].copy_(allv[q][a:b])
        if v.is_cuda:
            torch.cuda.synchronize()
        self._done()
        self.collectives += 1

    def or_allreduce(self, words, or_into, window=1 << 26):
        if words.numel() == 0:
            return
        allv = self._swap(words)
        acc = words.clone()
        for q in range(self.world):
            if q != self.rank:
                or_into(acc, allv[q])
        if acc.is_cuda:
            torch.cuda.synchronize()
        self.hub.barrier.wait()                                 # everybody has read everybody's partial filter
        words.copy_(acc)
        if acc.is_cuda:
            torch.cuda.synchronize()
        self._done()

    def barrier(self):
        self.hub.barrier.wait()


def run_threads(world, fn):
    """fn(rank, comm) on `world` threads; returns the results in rank order, re-raises the first failure"""
    hub = Hub(world)
    res, err = [None] * world, []

    def body(r):
        try:
            res[r] = fn(r, ThreadComm(hub, r))
        except BaseException as e:  # noqa: BLE001
            err.append(e)
            hub.barrier.abort()                                 # the others wait in a collective: let them out

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    real = [e for e in err if not isinstance(e, threading.BrokenBarrierError)]
    if real or err:
        raise (real or err)[0]
    return res
