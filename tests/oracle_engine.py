"""The engine interface of kmcex_amd.dist (count_classes / shard_begin / classify / ring_round / ... / complete) on top of
the CPU oracle, with CPU torch tensors.  Lets the tests run the SAME multi-rank orchestration (routing plan, ring of
arrays, OR-merge of partial filters, survivor gather) over gloo without a GPU and compare the result with the
sequential build.  Test infrastructure only."""
import ctypes as C

import numpy as np
import torch

import oracle_lib as O

BUCKET = 1 << 18
HDR = 8


def _np(t):
    return t.numpy()


class _Local:
    pass


class OracleEngine:
    def __init__(self, ci, cs, nh, nb):
        self.L = O.lib()
        L = self.L
        L.kmo_shard_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]
        L.kmo_shard_classify.restype = C.c_uint64
        L.kmo_shard_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.kmo_ring_insert.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.kmo_shard_complete.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        self.o = O.OracleModel(ci, cs, nh, nb)
        self.ci, self.cs, self.nh, self.nb, self.bf_num = ci, cs, nh, nb, (1 if ci == 1 else 3)
        self.device = torch.device("cpu")

    # ---- pass 1 / sizes
    def count_classes(self, counts):
        c = _np(counts).view(np.uint32)
        assert c.size == 0 or (c.min() >= self.ci and c.max() <= self.cs)
        return [int((c == self.ci + i).sum()) if i < self.bf_num else 0 for i in range(3)]

    def shard_begin(self, k, n_bf, n_total, rank, world):
        self.k, self.W = k, (k + 31) // 32
        arr = (C.c_uint64 * 3)(*[int(x) for x in n_bf])
        assert self.L.kmo_shard_begin(self.o.h, k, arr, n_total) == 0
        self.rest_k, self.rest_c = [], []
        self.stale = {}                                             # list -> (k-mer words, count) of slot 0 when it was retired (quirk Q1)
        self.extra_attempts = 0

    def classify(self, kmers, counts):
        n = counts.numel()
        km = np.ascontiguousarray(_np(kmers)).view(np.uint64)
        cn = np.ascontiguousarray(_np(counts)).view(np.uint32)
        ok, oc = np.zeros(max(n, 1) * self.W, dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint32)
        got = self.L.kmo_shard_classify(self.o.h, km.ctypes.data, cn.ctypes.data, n, ok.ctypes.data, oc.ctypes.data)
        assert got != (1 << 64) - 1
        tk = torch.from_numpy(ok[: got * self.W].view(np.int64).copy())
        return (tk.view(got, self.W) if self.W > 1 else tk), torch.from_numpy(oc[:got].view(np.int32).copy())

    # ---- the ring
    def new_message(self):
        return torch.zeros(HDR + BUCKET * self.W + BUCKET // 2, dtype=torch.int64)

    def _unpack(self, msg):
        a = _np(msg)
        n = int(a[0])
        km = a[HDR: HDR + BUCKET * self.W].view(np.uint64)[: n * self.W]
        cn = a[HDR + BUCKET * self.W:].view(np.uint32)[:n]
        return n, km, cn

    def ring_round(self, t, lists):
        for l in lists:
            i = l["list"]
            if l.get("n") is not None:
                n = l["n"]
                km = np.ascontiguousarray(_np(l["kmers"])).view(np.uint64).reshape(-1)
                cn = np.ascontiguousarray(_np(l["counts"])).view(np.uint32)
            else:
                n, km, cn = self._unpack(l["msg"])
            ok, oc = np.zeros(max(n, 1) * self.W, dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint32)
            left = 0
            if n > 0:
                left = self.L.kmo_ring_insert(self.o.h, (i + t) % self.nb, np.ascontiguousarray(km).ctypes.data, np.ascontiguousarray(cn).ctypes.data, n, ok.ctypes.data, oc.ctypes.data)
                assert left >= 0
            if l.get("out") is not None:
                a = _np(l["out"])
                a[0] = left
                a[HDR: HDR + BUCKET * self.W].view(np.uint64)[: left * self.W] = ok[: left * self.W]
                a[HDR + BUCKET * self.W:].view(np.uint32)[:left] = oc[:left]
            else:                                                   # kmodel.hpp:567-571 + what slot 0 keeps for the final block
                self.rest_k.append(ok[: left * self.W].copy())
                self.rest_c.append(oc[:left].copy())
                self.stale[i] = (ok[: self.W].copy(), int(oc[0])) if left > 0 else None

    def stale_dup(self, first_unused_row):
        for i in range(first_unused_row, self.nb):
            if self.stale.get(i):
                km, c = self.stale[i]
                self.rest_k.append(km.copy())
                self.rest_c.append(np.array([c], dtype=np.uint32))
                self.extra_attempts += self.nb - 1                  # it is offered to the nb-1 other arrays and fails on each

    def local(self):
        st = _Local()
        so = self.o.stats()
        st.attempts, st.successes = int(so.attempts) + self.extra_attempts, int(so.successes)
        st.fast_commits = st.contended = st.finisher_iters = 0
        km = np.concatenate(self.rest_k) if self.rest_k else np.zeros(0, dtype=np.uint64)
        cn = np.concatenate(self.rest_c) if self.rest_c else np.zeros(0, dtype=np.uint32)
        st.rest_entries = len(cn)
        tk = torch.from_numpy(km.view(np.int64))
        return st, (tk.view(len(cn), self.W) if self.W > 1 else tk), torch.from_numpy(cn.view(np.int32))

    # ---- merge
    def _bytes(self, ptr, n):
        if not n:
            return torch.zeros(0, dtype=torch.uint8)
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(n),)))     # shares the oracle's memory

    def view(self, which, index=0):
        st = self.o.stats()
        if which == "bf":
            return self._bytes(self.L.kmo_bf(self.o.h, index), st.byte_bf[index])
        if which == "bf_back":
            return self._bytes(self.L.kmo_bf_back(self.o.h, index), st.byte_bf_back[index])
        assert which == "km_back"
        return self._bytes(self.L.kmo_km_back(self.o.h), st.byte_km_back)

    def array_views(self, a):
        st = self.o.stats()
        return [self._bytes(self.L.kmo_value_array(self.o.h, a), st.km_byte_size), self._bytes(self.L.kmo_tag_array(self.o.h, a), st.km_byte_size)]

    def or_into(self, dst, src):
        np.bitwise_or(_np(dst), _np(src), out=_np(dst))

    def complete(self, rest_kmers, rest_counts, totals):
        km = np.ascontiguousarray(_np(rest_kmers)).view(np.uint64)
        cn = np.ascontiguousarray(_np(rest_counts)).view(np.int32)
        assert self.L.kmo_shard_complete(self.o.h, km.ctypes.data, cn.ctypes.data, len(cn), int(totals.attempts), int(totals.successes)) == 0
