"""One KModel across several GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on MI355X;
"gloo" for the CPU tests and for rehearsing N ranks on a one-GPU box).

What the reference does with OpenMP threads inside ONE process (SURVEY.md §2 a-d, §8e) maps onto the ranks like this:

* **coupled arrays = a ring** -- the reference's insert is the rotation ``insert_array(buff[i], (i + t) % n_thread, ...)``
  (``kmodel.hpp:560-565``): thread ``i`` walks buffer ``i`` against array ``(i + t) % nb``, barrier, next ``t``.  Here the
  arrays are owned whole (``owner_of_array``): round ``t`` of a block runs on the rank that owns the array, then the
  survivors of the list travel, in list order, to the owner of the next array (RCCL send/recv, one xGMI hop).
  Round 0 needs buffer ``i`` of every block on the owner of array ``i``: the k-mers are routed there by ONE
  **all-to-all** from the ranks that listed them (``plan_routing``).
* **Bloom filters, back filters, km_back** are order-free (``set_bit`` is an OR, ``kmodel.hpp:576-581``): every rank fills
  partial filters from what it handles and the partials are merged by position range -- an all-to-all of ranges, an OR
  on the owner, an all-gather (``or_allreduce``).  That moves 1.7 B per Bloom k-mer and 0.3 B per coupled k-mer
  instead of 8 B per hashed position.
* **queries** shard by batch over read-only replicas (``query_replicas``): after the merge every rank holds the whole
  model (47 GB at 10^10 k-mers, far below 288 GB).

The per-rank compute is behind an *engine* (``DeviceEngine`` = libkmx.so through the C ABI; the tests also drive the same
orchestration with a CPU engine made of the oracle, which is how the N>1 protocol is checked without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

BUCKET = 1 << 18            # kmodel.hpp:276 bucket_size


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched bare."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def stream_seeds(rank: int):
    """Per-rank generator seeds: rank 0 is the single-GPU workload (seed_k=1, seed_c=2)."""
    return 1 + 1000003 * rank, 2 + 1000003 * rank


def split_batch(n: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of an n-element batch for `rank`; slices differ by at most one element."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_job(seconds, units, device="cpu"):
    """Whole-job figures from per-rank ones: MAX over ranks of every time, SUM over ranks of every unit count."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(seconds), list(units)
    t = torch.tensor(list(seconds), dtype=torch.float64, device=device)
    u = torch.tensor(list(units), dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return t.tolist(), u.tolist()


def gather_slices(local: torch.Tensor, n_total: int, world: int, rank: int, comm: "Comm | None" = None):
    """All-gather the per-rank answers of a batch split with `split_batch` back into batch order (ragged slices)."""
    if world == 1:
        return local
    sizes = [split_batch(n_total, world, r) for r in range(world)]
    if comm is not None:
        return comm.all_gather_v(local, [hi - lo for lo, hi in sizes])
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(width, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)])


# ------------------------------------------------------------------------------------------------ who owns what
def owner_of_array(a: int, nb: int, world: int) -> int:
    """Arrays are owned whole, in contiguous runs, by the first min(world, nb) ranks: consecutive arrays mostly share a
    rank, so a list changes GPUs only where a run ends (the rotation visits the arrays in order, kmodel.hpp:563)."""
    return a * min(world, nb) // nb


def _segments(off: int, c: int, nb: int):
    """The range [off, off + c) of the coupled-array stream cut at buffer boundaries: (lo, hi, buffer index) rows.
    Element g of the stream sits in buffer (g >> 18) % nb of block g // (nb << 18) (push_to_array, kmodel.hpp:508-513)."""
    if c <= 0:
        return np.zeros((0, 3), dtype=np.int64)
    first, last = off // BUCKET, (off + c - 1) // BUCKET
    seg = np.arange(first, last + 1, dtype=np.int64)
    lo = np.maximum(seg * BUCKET, off)
    hi = np.minimum((seg + 1) * BUCKET, off + c)
    return np.stack([lo, hi, seg % nb], axis=1)


def plan_routing(counts_per_rank, nb: int, world: int, rank: int, own=None):
    """All-to-all plan that brings every buffer of the coupled-array stream to the rank owning the array it meets first
    (`own[i]`: the rank buffer i of every block goes to; default: the ring's owner of array i).

    `counts_per_rank[q]` = coupled-array k-mers rank q holds (its listing slice, in listing order; slices are in rank
    order).  Returns (send_slices, send_splits, recv_splits): local [lo, hi) slices in send order (grouped by
    destination, ascending stream position inside a destination) and the per-rank split sizes.  Received pieces, taken
    in source-rank order, are this rank's buffers in ascending stream position."""
    offs = np.concatenate([[0], np.cumsum(np.asarray(counts_per_rank, dtype=np.int64))])
    own = np.array([owner_of_array(a, nb, world) for a in range(nb)] if own is None else own, dtype=np.int64)
    recv_splits = []
    for q in range(world):
        seg = _segments(int(offs[q]), int(counts_per_rank[q]), nb)
        mine = own[seg[:, 2]] == rank
        recv_splits.append(int((seg[mine, 1] - seg[mine, 0]).sum()))
    seg = _segments(int(offs[rank]), int(counts_per_rank[rank]), nb)
    dest = own[seg[:, 2]]
    send_slices, send_splits = [], []
    for d in range(world):
        rows = seg[dest == d]
        send_slices += [(int(lo - offs[rank]), int(hi - offs[rank])) for lo, hi, _ in rows]
        send_splits.append(int((rows[:, 1] - rows[:, 0]).sum()))
    return send_slices, send_splits, recv_splits


def list_length(n_km: int, nb: int, b: int, i: int) -> int:
    """entries of buffer i in block b (push_to_array / push_last_to_array, kmodel.hpp:508-527)"""
    return int(min(max(n_km - (b * nb + i) * BUCKET, 0), BUCKET))


MSG_HDR = 8                                                 # 64-bit words before the k-mers of a ring message (kmx_types.h KMX_MSG_HDR)


def ring_msg_pieces(msg: torch.Tensor, n: int, W: int):
    """the words of a ring message that n survivors occupy: header + n k-mers, and the n 32-bit counts behind the k-mer area"""
    n = max(0, min(int(n), BUCKET))
    c0 = MSG_HDR + BUCKET * W
    return [msg[:MSG_HDR + n * W], msg[c0:c0 + (n + 1) // 2]]


# ------------------------------------------------------------------------------------------------ the exchange
class Comm:
    """The collectives of the sharded build on top of torch.distributed.  Backend "nccl" (RCCL over xGMI) moves device
    tensors directly; with "gloo" (CPU tests, rehearsal of N ranks on one GPU) device tensors are staged through the host.
    """

    def __init__(self, group=None, shortcut=True):
        self.group = group
        self.shortcut = shortcut                            # False: a single rank still goes through the backend (tests of the RCCL path)
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.staged = self.on and dist.get_backend(group) != "nccl"
        self.bytes_sent = 0                                 # payload this rank handed to the backend (reporting)
        self.collectives = 0                                # data all-to-alls this rank took part in

    def _wire(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if (self.staged and t.is_cuda) else t

    def all_reduce_ints(self, values, device, op=None):
        if self.world == 1 and self.shortcut:
            return [int(v) for v in values]
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device="cpu" if self.staged else device)
        dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=self.group)
        return [int(v) for v in t.tolist()]

    def all_gather_ints(self, value: int, device):
        if self.world == 1 and self.shortcut:
            return [int(value)]
        t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if self.staged else device)
        out = torch.empty(self.world, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return [int(v) for v in out.tolist()]

    def all_to_all_ints(self, values, device):
        """values[q] goes to rank q; returns what every rank sent here (the split sizes of a ragged all-to-all)"""
        if self.world == 1 and self.shortcut:
            return [int(v) for v in values]
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device="cpu" if self.staged else device)
        out = torch.empty_like(t)
        dist.all_to_all_single(out, t, group=self.group)
        return [int(v) for v in out.tolist()]

    def all_to_all_v(self, send: torch.Tensor, send_splits, recv_splits) -> torch.Tensor:
        """rows of `send` grouped by destination -> rows received, grouped by source"""
        if self.world == 1 and self.shortcut:
            return send
        n_recv = int(sum(recv_splits))
        w = self._wire(send.contiguous())
        out = torch.empty((n_recv,) + tuple(send.shape[1:]), dtype=send.dtype, device=w.device)
        dist.all_to_all_single(out, w, [int(x) for x in recv_splits], [int(x) for x in send_splits], group=self.group)
        self.collectives += 1
        self.bytes_sent += (int(sum(send_splits)) - int(send_splits[self.rank])) * send.element_size() * int(np.prod(send.shape[1:], dtype=np.int64))
        return out.to(send.device) if out.device != send.device else out

    def all_to_all_fixed(self, send: torch.Tensor) -> torch.Tensor:
        """send[q] (equal pieces, one per rank) goes to rank q; returns what every rank sent here, same shape -- no split size
        comes from the host"""
        if self.world == 1 and self.shortcut:
            return send
        w = self._wire(send.contiguous())
        out = torch.empty_like(w)
        dist.all_to_all_single(out, w, group=self.group)
        self.collectives += 1
        self.bytes_sent += send.numel() * send.element_size() * (self.world - 1) // self.world
        return out.to(send.device) if out.device != send.device else out

    def exchange(self, sends, recvs):
        """point-to-point hand-offs of one ring step: sends = [(tensor, dst rank)], recvs = [(tensor, src rank)]"""
        if not sends and not recvs:
            return
        stage_out = [(self._wire(t), d) for t, d in sends]
        stage_in = [(torch.empty(t.shape, dtype=t.dtype, device="cpu") if (self.staged and t.is_cuda) else t, s) for t, s in recvs]
        ops = [dist.P2POp(dist.isend, t, d, group=self.group) for t, d in stage_out] + [dist.P2POp(dist.irecv, t, s, group=self.group) for t, s in stage_in]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for (dst, _), (st, _) in zip(recvs, stage_in):
            if st is not dst:
                dst.copy_(st)
        self.bytes_sent += sum(t.numel() * t.element_size() for t, _ in sends)

    def exchange_counted(self, sends, recvs, pieces):
        """`exchange` for messages whose word 0 says how much of them is in use: the fill counts travel first (one word
        per message), then only `pieces(msg, n)` -> [contiguous views] of every message.  Reading word 0 waits for the
        kernels that wrote it, so this form is for the staged transport, which waits for them anyway; over RCCL the ring
        step stays free of host synchronisation and ships whole buffers."""
        if not sends and not recvs:
            return
        n_out = [int(t[0]) for t, _ in sends]
        heads = [torch.empty(1, dtype=torch.int64) for _ in recvs]
        self.exchange([(torch.tensor([n], dtype=torch.int64), d) for n, (_, d) in zip(n_out, sends)], [(h, s) for h, (_, s) in zip(heads, recvs)])
        self.exchange([(v, d) for n, (t, d) in zip(n_out, sends) for v in pieces(t, n)],
                      [(v, s) for h, (t, s) in zip(heads, recvs) for v in pieces(t, int(h[0]))])

    def broadcast(self, t: torch.Tensor, src: int):
        if self.world == 1 and self.shortcut:
            return
        if self.staged and t.is_cuda:
            h = t.cpu()
            dist.broadcast(h, src, group=self.group)
            if self.rank != src:
                t.copy_(h)
        else:
            dist.broadcast(t, src, group=self.group)
        if self.rank == src:
            self.bytes_sent += t.numel() * t.element_size()

    def all_gather_v(self, local: torch.Tensor, counts) -> torch.Tensor:
        """ragged all-gather of rows: rank q contributes counts[q] rows; result in rank order"""
        if self.world == 1 and self.shortcut:
            return local
        width = int(max(counts))
        pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        w = self._wire(pad)
        out = torch.empty((self.world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=w.device)
        dist.all_gather_into_tensor(out, w, group=self.group)
        self.bytes_sent += pad.numel() * pad.element_size() * (self.world - 1)
        out = out.to(local.device) if out.device != local.device else out
        return torch.cat([out[q * width: q * width + int(counts[q])] for q in range(self.world)])

    def all_gather_ranges(self, v: torch.Tensor, lo, window: int = 1 << 24):
        """every rank holds its range v[lo[rank]:lo[rank + 1]] of `v`; on return every rank holds all of `v` -- in place, window
        by window (the ranges of a position-range partition differ by at most one element, so a window is one all_gather_into_tensor
        of equal pieces; the temporaries are world x window elements whatever the array's size is)"""
        if self.world == 1 and self.shortcut:
            return
        width = max(int(lo[q + 1]) - int(lo[q]) for q in range(self.world))
        for off in range(0, width, window):
            w = min(window, width - off)
            mine = v[int(lo[self.rank]) + off: min(int(lo[self.rank]) + off + w, int(lo[self.rank + 1]))]
            piece = mine if mine.numel() == w else torch.cat([mine, torch.zeros(w - mine.numel(), dtype=v.dtype, device=v.device)])
            wire = self._wire(piece.contiguous())
            out = torch.empty(self.world * w, dtype=v.dtype, device=wire.device)
            dist.all_gather_into_tensor(out, wire, group=self.group)
            self.collectives += 1
            self.bytes_sent += piece.numel() * piece.element_size() * (self.world - 1)
            for q in range(self.world):
                a, b = int(lo[q]) + off, min(int(lo[q]) + off + w, int(lo[q + 1]))
                if q != self.rank and b > a:
                    v[a:b].copy_(out[q * w: q * w + (b - a)])

    def or_allreduce(self, words: torch.Tensor, or_into, window: int = 1 << 26):
        """Bitwise OR of `words` (int32, same length on every rank) over the ranks, in place, by position range: an
        all-to-all brings range r of every rank's partial filter to rank r, `or_into(dst, src)` merges them there, an
        all-gather hands every rank the merged filter (RCCL has no OR reduction).  Big slabs go window by window (2^26
        words = 256 MB at a time), so the temporaries stay a fixed size whatever the model's."""
        if (self.world == 1 and self.shortcut) or words.numel() == 0:
            return
        P = self.world
        for lo in range(0, words.numel(), window):
            part = words[lo:lo + window]
            n = part.numel()
            chunk = -(-n // P)
            if chunk * P == n:
                pad = part                                           # the all-to-all only reads it
            else:
                pad = torch.zeros(chunk * P, dtype=words.dtype, device=words.device)
                pad[:n] = part
            w = self._wire(pad)
            got = torch.empty_like(w)
            dist.all_to_all_single(got, w, group=self.group)
            got = got.to(words.device) if got.device != words.device else got
            acc = got[:chunk]
            for q in range(1, P):
                or_into(acc, got[q * chunk: (q + 1) * chunk])
            aw = self._wire(acc.contiguous())
            full = pad if (pad is not part and not self.staged) else torch.empty(chunk * P, dtype=words.dtype, device=aw.device)
            dist.all_gather_into_tensor(full, aw, group=self.group)
            part.copy_(full[:n])
            self.bytes_sent += 2 * chunk * (P - 1) * words.element_size()
            del pad, got, acc, full

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)


# ------------------------------------------------------------------------------------------------ the GPU engine
class _DevMem:
    """Zero-copy torch view of device memory that libkmx owns (filters, arrays, survivor lists)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}


def dev_tensor(ptr: int, n: int, dtype: torch.dtype, device) -> torch.Tensor:
    if n == 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DevMem(ptr, n, {torch.int32: "<i4", torch.int64: "<i8"}[dtype]), device=device)


class DeviceEngine:
    """This rank's share of the model in HBM: libkmx.so through the C ABI (include/kmx.h, "ONE model built by several GPUs").
    Tensors are int64 (packed k-mers, uint64 bit patterns) and int32 (counts, filter words) on `device`."""

    def __init__(self, model, device):
        self.m, self.device = model, device
        model.set_stream(torch.cuda.current_stream(device).cuda_stream)     # kernels and collectives order on ONE stream

    def count_classes(self, counts):
        return self.m.count_classes_dev(counts.data_ptr(), counts.numel())

    def shard_begin(self, k, n_bf, n_total, rank, world):
        self.k, self.W = k, (k + 31) // 32
        self.m.shard_begin(k, n_bf, n_total, rank, world)

    def classify(self, kmers, counts):
        n = counts.numel()
        out_k = torch.empty_like(kmers)
        out_c = torch.empty_like(counts)
        n_out = self.m.shard_classify_dev(kmers.data_ptr(), counts.data_ptr(), n, out_k.data_ptr(), out_c.data_ptr())
        return out_k[:n_out], out_c[:n_out]

    def new_message(self):
        return torch.zeros(self.m.ring_msg_bytes(self.k) // 8, dtype=torch.int64, device=self.device)

    def ring_round(self, t, lists):
        """lists: dicts {list, n (or None), kmers, counts, msg, out}; tensors or None"""
        ptr = lambda x: 0 if x is None else x.data_ptr()                                   # noqa: E731
        self.m.ring_round_dev(t, [(l["list"], -1 if l.get("n") is None else l["n"], ptr(l.get("kmers")), ptr(l.get("counts")), ptr(l.get("msg")), ptr(l.get("out"))) for l in lists])

    def stale_dup(self, first_unused_row):
        self.m.ring_stale_dup_dev(first_unused_row)

    # ---- position-range partition (include/kmx.h kmx_range_*): words by destination rank in device "send regions"
    def range_begin(self, k, n_bf, n_total, rank, world):
        self.k, self.W = k, (k + 31) // 32
        self.m.range_begin(k, n_bf, n_total, rank, world)
        self._world = world
        p, self._cap, self.cell_lo = self.m.range_buffers()
        self._send = dev_tensor(p, self._cap * world, torch.int64, self.device)

    def _regions(self, counts):
        """what the last emit / resolve left for every rank, concatenated in rank order (the order the verdicts come back in)"""
        parts = [self._send[q * self._cap: q * self._cap + c] for q, c in enumerate(counts) if c]
        if len(parts) == 1:
            return parts[0]                                     # one destination (one rank, or nothing for the others): the region itself, no copy
        return torch.cat(parts) if parts else self._send[:0]

    def range_emit(self, t, lists):
        """-> the regions back to back, words per destination, commit words among them (in front of each region)"""
        counts, commits = self.m.range_emit_dev(t, [(l["list"], l["n"], l["kmers"].data_ptr() if l["n"] else 0, l["counts"].data_ptr() if l["n"] else 0) for l in lists])
        return self._regions(counts), counts, commits

    def range_verdict(self, t, words, totals, commits):
        """words: what the senders left for this rank, back to back (totals[s] words of sender s, commits[s] commit words in front)"""
        ver = torch.empty(words.numel(), dtype=torch.uint8, device=self.device)
        if words.numel():
            self.m.range_verdict_dev(t, words.data_ptr(), totals, commits, ver.data_ptr())
        return ver

    def range_resolve(self, t, verdicts):
        """the winners' commits go to the front of the regions; the next emit appends its triples behind them"""
        self.m.range_resolve_dev(t, verdicts.data_ptr() if verdicts.numel() else 0)

    def range_flush(self):
        counts = self.m.range_flush_dev()
        return self._regions(counts), counts

    # fixed-size messages (kmx_range_inband): a region = [header | capx words]; no count ever reaches the host
    def range_inband(self):
        p, self._rw, self._capx = self.m.range_inband()
        self._msg = dev_tensor(p, self._rw * self._world, torch.int64, self.device).view(self._world, self._rw)

    def range_emit_inband(self, t, lists):
        self.m.range_emit_nowait_dev(t, [(l["list"], l["n"], l["kmers"].data_ptr() if l["n"] else 0, l["counts"].data_ptr() if l["n"] else 0) for l in lists])
        return self._msg

    def range_verdict_inband(self, t, recv):
        ver = torch.empty((self._world, self._capx), dtype=torch.uint8, device=self.device)
        self.m.range_verdict_inband_dev(t, recv.data_ptr(), self._world, ver.data_ptr())
        return ver

    def range_flush_inband(self):
        self.m.range_flush_nowait_dev()
        return self._msg

    def range_commit_inband(self, recv):
        self.m.range_commit_inband_dev(recv.data_ptr(), self._world)

    def range_commit(self, commits):
        if commits.numel():
            self.m.range_commit_dev(commits.data_ptr(), commits.numel())

    def array_ranges(self, a):
        return [(self.view("cells", a), self.cell_lo)]

    def local(self):
        st, pk, pc = self.m.shard_local()
        n = int(st.rest_entries)
        km = dev_tensor(pk, n * self.W, torch.int64, self.device)
        return st, (km.view(n, self.W) if self.W > 1 else km), dev_tensor(pc, n, torch.int32, self.device)

    def view(self, which, index=0):
        p, nbytes = self.m.dev_view(which, index)
        return dev_tensor(p, nbytes // 4, torch.int32, self.device)

    def array_views(self, a):
        return [self.view("cells", a)]                              # value and tag bits interleaved: one tensor per array

    def or_into(self, dst, src):
        self.m.or_words_dev(dst.data_ptr(), src.data_ptr(), dst.numel())

    def complete(self, rest_kmers, rest_counts, totals):
        rest_kmers, rest_counts = rest_kmers.contiguous(), rest_counts.contiguous()
        self.m.shard_complete(rest_kmers.data_ptr() if rest_counts.numel() else 0, rest_counts.data_ptr() if rest_counts.numel() else 0, rest_counts.numel(), totals)


# ------------------------------------------------------------------------------------------------ the sharded build
STAT_FIELDS = ("attempts", "successes", "fast_commits", "contended", "finisher_iters")


def build_sharded(eng, comm: Comm, k: int, nb: int, bf_num: int, kmers: torch.Tensor, counts: torch.Tensor, n_total: int | None = None, partition: str = "ring"):
    """KModel::init (kmodel.hpp:57-86) of ONE model by `comm.world` ranks.  `kmers` / `counts`: this rank's contiguous slice
    of the listing (slices in rank order = listing order).  On return every rank holds the whole model.
    `partition`: "ring" -- arrays owned whole, lists travel (module docstring); "range" -- every array cut by position
    range, k-mers stay, commits + triples out and verdicts back travel by all-to-all (`build_range_sharded`).

    Returns a dict of figures (n_km, blocks, bytes this rank sent)."""
    if partition == "range":
        return build_range_sharded(eng, comm, k, nb, bf_num, kmers, counts, n_total)
    if partition != "ring":
        raise ValueError(f"partition {partition!r}: ring or range")
    rank, world, dev = comm.rank, comm.world, counts.device
    sent0 = comm.bytes_sent
    # pass 1 (get_km_kmer_count, kmodel.hpp:423-434): class histogram of the slice, summed over the ranks
    local_hist = eng.count_classes(counts)
    tot = comm.all_reduce_ints(local_hist + [counts.numel()], dev)
    n_bf, n_all = tot[:3], tot[3]
    eng.shard_begin(k, n_bf, n_all if n_total is None else n_total, rank, world)
    # pass 2 front end on the slice: partial Bloom/back filters + this rank's coupled-array k-mers in listing order
    km_loc, cnt_loc = eng.classify(kmers, counts)
    per_rank = comm.all_gather_ints(cnt_loc.shape[0], dev)
    n_km = int(sum(per_rank))
    # all-to-all: every buffer of the stream goes to the owner of the array it meets in round 0
    send_slices, send_splits, recv_splits = plan_routing(per_rank, nb, world, rank)
    if world > 1:
        pick = lambda t: torch.cat([t[lo:hi] for lo, hi in send_slices]) if send_slices else t[:0]      # noqa: E731
        km_mine = comm.all_to_all_v(pick(km_loc), send_splits, recv_splits)
        cnt_mine = comm.all_to_all_v(pick(cnt_loc), send_splits, recv_splits)
    else:
        km_mine, cnt_mine = km_loc, cnt_loc
    del km_loc, cnt_loc
    own = [owner_of_array(a, nb, world) for a in range(nb)]
    blk = nb * BUCKET
    n_blocks = -(-n_km // blk)
    msgs = {}

    def msg(i, parity):
        if (i, parity) not in msgs:
            msgs[(i, parity)] = eng.new_message()
        return msgs[(i, parity)]

    pos = 0                                                     # read position in this rank's routed stream
    for b in range(n_blocks):
        n_in_block = min(blk, n_km - b * blk)
        if n_in_block < blk and b > 0:                          # final partial block: quirk Q1 (kmodel.hpp:520-527)
            row = (n_in_block - 1) // BUCKET
            if row + 1 < nb:
                eng.stale_dup(row + 1)
        for t in range(nb):
            lists, sends, recvs = [], [], []
            for i in range(nb):
                n_i = list_length(n_km, nb, b, i)
                a = (i + t) % nb                                # kmodel.hpp:563
                if n_i == 0:
                    continue
                if own[a] == rank:
                    ent = {"list": i, "out": msg(i, (t + 1) & 1) if t + 1 < nb else None}
                    if t == 0:
                        ent.update(n=n_i, kmers=km_mine[pos:pos + n_i], counts=cnt_mine[pos:pos + n_i])
                        pos += n_i
                    else:
                        ent.update(msg=msg(i, t & 1))
                    lists.append(ent)
                    if t + 1 < nb and own[(a + 1) % nb] != rank:
                        sends.append((msg(i, (t + 1) & 1), own[(a + 1) % nb]))
                elif t + 1 < nb and own[(a + 1) % nb] == rank:
                    recvs.append((msg(i, (t + 1) & 1), own[a]))
            if lists:
                eng.ring_round(t, lists)
            if comm.staged:
                comm.exchange_counted(sends, recvs, lambda m, n: ring_msg_pieces(m, n, (k + 31) // 32))
            else:
                comm.exchange(sends, recvs)
    # merge: survivors -> every rank; filters OR-merged by range; every array from its owner
    st, rest_km, rest_cnt = eng.local()
    rest_counts = comm.all_gather_ints(int(st.rest_entries), dev)
    rest_km_all = comm.all_gather_v(rest_km, rest_counts)
    rest_cnt_all = comm.all_gather_v(rest_cnt, rest_counts)
    sums = comm.all_reduce_ints([getattr(st, f) for f in STAT_FIELDS], dev)
    for f, v in zip(STAT_FIELDS, sums):
        setattr(st, f, v)
    st.blocks, st.rounds = n_blocks, n_blocks * nb
    if world > 1:
        for i in range(bf_num):
            comm.or_allreduce(eng.view("bf", i), eng.or_into)
            comm.or_allreduce(eng.view("bf_back", i), eng.or_into)
        comm.or_allreduce(eng.view("km_back"), eng.or_into)
        for a in range(nb):
            for v in eng.array_views(a):
                comm.broadcast(v, own[a])
    eng.complete(rest_km_all, rest_cnt_all, st)
    return {"n_km": n_km, "blocks": n_blocks, "bytes_sent": comm.bytes_sent - sent0, "arrays_owned": [a for a in range(nb) if own[a] == rank]}


def build_range_sharded(eng, comm: Comm, k: int, nb: int, bf_num: int, kmers: torch.Tensor, counts: torch.Tensor, n_total: int | None = None, messages: str | None = None):
    """The north star's partition: "shard the bit arrays by hash-range across up to 8 GPUs with an RCCL all-to-all".  Rank q
    owns the cells [cell_lo[q], cell_lo[q+1]) of EVERY array; list i of a block lives on rank i % world, which hashes its
    k-mers, keeps the list order and reorders locally -- k-mers are routed once (the same all-to-all as the ring's, to the
    list's rank) and never move again.  A round (list i against array (i + t) % nb, kmodel.hpp:560-565) is two all-to-alls:
    64-bit words to the range owners -- the winners' commits of the round before, then this round's triples -- and one verdict
    byte per word back (`range_kernels.h` has the kernels and why the outcome is the sequential one); the last round's commits
    are flushed by one more exchange at the end.  Every rank works in every round, whatever nb is.

    messages = "fixed" (the default): a region travels as [header | capx words] -- the counts are IN BAND, both all-to-alls of a
    round have equal splits, and NO number of a round reaches the host: 2 collectives, 0 host waits per round.  capx is the mean
    of a round's fullest exchange + 25 % + 8192 words; should a region ever overflow (uniformly hashed positions do not), the
    words that did not fit were dropped, every rank learns it with the final statistics, and the build is repeated with
    messages = "counted": split sizes from the host, a ragged all-to-all each way -- 3 collectives and a host wait per round."""
    if messages is None:
        messages = "counted" if os.environ.get("KMX_RANGE_MESSAGES") == "counted" else "fixed"
    if messages not in ("fixed", "counted"):
        raise ValueError(f"messages {messages!r}: fixed or counted")
    fixed = messages == "fixed" and hasattr(eng, "range_inband")
    rank, world, dev = comm.rank, comm.world, counts.device
    sent0, coll0 = comm.bytes_sent, comm.collectives
    local_hist = eng.count_classes(counts)
    tot = comm.all_reduce_ints(local_hist + [counts.numel()], dev)
    n_bf, n_all = tot[:3], tot[3]
    eng.range_begin(k, n_bf, n_all if n_total is None else n_total, rank, world)
    if fixed:
        eng.range_inband()
    km_loc, cnt_loc = eng.classify(kmers, counts)
    per_rank = comm.all_gather_ints(cnt_loc.shape[0], dev)
    n_km = int(sum(per_rank))
    own = [i % world for i in range(nb)]                        # buffer i of every block -> the rank that holds list i
    send_slices, send_splits, recv_splits = plan_routing(per_rank, nb, world, rank, own)
    if world > 1:
        pick = lambda t: torch.cat([t[lo:hi] for lo, hi in send_slices]) if send_slices else t[:0]      # noqa: E731
        km_mine = comm.all_to_all_v(pick(km_loc), send_splits, recv_splits)
        cnt_mine = comm.all_to_all_v(pick(cnt_loc), send_splits, recv_splits)
    else:
        km_mine, cnt_mine = km_loc, cnt_loc
    del km_loc, cnt_loc
    blk = nb * BUCKET
    n_blocks = -(-n_km // blk)
    pos = 0
    for b in range(n_blocks):
        n_in_block = min(blk, n_km - b * blk)
        if n_in_block < blk and b > 0:                          # final partial block: quirk Q1 (kmodel.hpp:520-527), on the rank that holds the list
            row = (n_in_block - 1) // BUCKET
            if row + 1 < nb:
                eng.stale_dup(row + 1)
        for t in range(nb):
            lists = []
            if t == 0:
                for i in range(rank, nb, world):
                    n_i = list_length(n_km, nb, b, i)
                    lists.append({"list": i, "n": n_i, "kmers": km_mine[pos:pos + n_i], "counts": cnt_mine[pos:pos + n_i]})
                    pos += n_i
            if fixed:
                got = comm.all_to_all_fixed(eng.range_emit_inband(t, lists))            # 1. [header | last round's commits + this round's triples] -> range owners
                back = comm.all_to_all_fixed(eng.range_verdict_inband(t, got))          # 2. commits applied, one verdict byte per word back
                eng.range_resolve(t, back)                                              # 3. winners decided; their commits go to the front of the regions
                continue
            words, out_counts, out_commits = eng.range_emit(t, lists)                   # 1. last round's commits + this round's triples -> range owners
            hdr = comm.all_to_all_ints([c | (cc << 32) for c, cc in zip(out_counts, out_commits)], dev)   # the regions' headers: words | commit words in front << 32
            in_counts, in_commits = [h & 0xFFFFFFFF for h in hdr], [h >> 32 for h in hdr]
            got = comm.all_to_all_v(words, out_counts, in_counts)
            ver = eng.range_verdict(t, got, in_counts, in_commits)                      # 2. commits applied, one verdict byte per word back, same order
            back = comm.all_to_all_v(ver, in_counts, out_counts)
            eng.range_resolve(t, back)                                                  # 3. winners decided; their commits go to the front of the regions
    if n_blocks:                                                                        # the last round's commits
        if fixed:
            eng.range_commit_inband(comm.all_to_all_fixed(eng.range_flush_inband()))
        else:
            commits, c_out = eng.range_flush()
            c_in = comm.all_to_all_ints(c_out, dev)
            eng.range_commit(comm.all_to_all_v(commits, c_out, c_in))
    st, rest_km, rest_cnt = eng.local()
    sums = comm.all_reduce_ints([getattr(st, f) for f in STAT_FIELDS] + [int(getattr(st, "reserved", 0) != 0)], dev)
    if fixed and sums[-1]:
        # a region dropped words on some rank: this build is void on every rank; once more, with counted messages (exact whatever the input)
        info = build_range_sharded(eng, comm, k, nb, bf_num, kmers, counts, n_total, messages="counted")
        info["fixed_messages_overflowed"] = True
        return info
    rest_counts = comm.all_gather_ints(int(st.rest_entries), dev)
    rest_km_all = comm.all_gather_v(rest_km, rest_counts)
    rest_cnt_all = comm.all_gather_v(rest_cnt, rest_counts)
    for f, v in zip(STAT_FIELDS, sums):
        setattr(st, f, v)
    st.blocks, st.rounds = n_blocks, n_blocks * nb
    if world > 1:
        for i in range(bf_num):
            comm.or_allreduce(eng.view("bf", i), eng.or_into)
            comm.or_allreduce(eng.view("bf_back", i), eng.or_into)
        comm.or_allreduce(eng.view("km_back"), eng.or_into)
        for a in range(nb):                                      # every rank's cell range of every array -> every rank
            for v, lo in eng.array_ranges(a):                    # (a view of the array and the ranks' bounds in its elements)
                comm.all_gather_ranges(v, lo)
    eng.complete(rest_km_all, rest_cnt_all, st)
    return {"n_km": n_km, "blocks": n_blocks, "bytes_sent": comm.bytes_sent - sent0, "collectives": comm.collectives - coll0,
            "cells_owned": [int(eng.cell_lo[rank]), int(eng.cell_lo[rank + 1])], "partition": "range", "messages": "fixed" if fixed else "counted"}


def query_replicas(model, comm: Comm, queries: torch.Tensor, k: int) -> torch.Tensor:
    """Batched kmer_to_occ (kmodel.hpp:90-98) against ONE model replicated on every rank: each rank answers its slice of
    the batch (the reference splits the vector over OpenMP threads the same way), the answers are gathered in batch order."""
    W = (k + 31) // 32
    n = queries.numel() // W
    lo, hi = split_batch(n, comm.world, comm.rank)
    out = torch.empty(hi - lo, dtype=torch.int32, device=queries.device)
    if hi > lo:
        q = queries.reshape(n, W)[lo:hi].contiguous()
        model.kmer_to_occ_dev(q.data_ptr(), hi - lo, out.data_ptr())
    return gather_slices(out, n, comm.world, comm.rank, comm)
