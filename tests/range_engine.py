"""The range-partition engine interface of kmcex_amd.dist.build_range_sharded (range_begin / range_emit / range_verdict /
range_resolve / range_commit / array_ranges) in numpy on top of the CPU oracle's arrays: lets the tests run the product's
orchestration of the north star's partition -- routing of buffer i to rank i % P, split-size exchange, the three all-to-alls of
a round, the all-gather of the cell ranges -- over gloo without a GPU.  k <= 32 (the vectorised hash).  Test infrastructure only."""
import os
import sys

import numpy as np
import torch

import oracle_lib as O
from oracle_engine import BUCKET, OracleEngine, _Local, _np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from range_shard_rehearsal import murmur64_packed, reorder     # noqa: E402  (the vectorised MurmurHash64A and reorder_buffer of the rehearsal)

M36 = (1 << 36) - 1


def _word(pos, want, i, x, j):
    return pos.astype(np.int64) | (want.astype(np.int64) << 36) | (np.int64(i) << 37) | (x.astype(np.int64) << 41) | (j.astype(np.int64) << 59)


class RangeOracleEngine(OracleEngine):
    def range_begin(self, k, n_bf, n_total, rank, world):
        assert k <= 32
        self.shard_begin(k, n_bf, n_total, rank, world)
        self.rank, self.world = rank, world
        so = self.o.stats()
        self.Lbits = int(so.km_byte_size) * 8
        ncells = (int(so.km_byte_size) + 1) // 2
        self.cell_lo = [q * ncells // world for q in range(world + 1)]
        self.seeds = [int(O.lib().kmo_hash_seed(i)) for i in range(128)]
        self.bin_of_occ, _ = O.occubin_table(self.cs + 1, self.nh)
        self.lists = {}                                             # list -> (kmers u64[n], counts u32[n]) in list order
        self.pending_w, self.pending_p = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)     # commits of the last round (words, positions)
        self.attempts = self.successes = self.n_contended = 0
        self.extra_attempts = 0

    def _owner(self, pos):
        return np.searchsorted(np.array(self.cell_lo[1:], dtype=np.int64), pos >> 4, side="right").astype(np.int64)

    def _bits(self, a):
        """tag / value arrays of array a as the oracle holds them (MSB-first bits, kmodel.hpp:576-588)"""
        so = self.o.stats()
        tag = np.ctypeslib.as_array(O.C.cast(self.L.kmo_tag_array(self.o.h, a), O.C.POINTER(O.C.c_uint8)), shape=(int(so.km_byte_size),))
        val = np.ctypeslib.as_array(O.C.cast(self.L.kmo_value_array(self.o.h, a), O.C.POINTER(O.C.c_uint8)), shape=(int(so.km_byte_size),))
        return tag, val

    @staticmethod
    def _get(arr, pos):
        return (arr[pos >> 3] >> (7 - (pos & 7)).astype(np.uint8)) & 1

    def _by_dest(self, words, pos):
        dest = self._owner(pos) if len(pos) else np.zeros(0, dtype=np.int64)
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=self.world).astype(np.int64).tolist()
        return words[order], order, counts

    def range_emit(self, t, lists):
        if t == 0:
            self.lists = {l["list"]: (np.ascontiguousarray(_np(l["kmers"])).view(np.uint64).reshape(-1)[: l["n"]].copy(),
                                      np.ascontiguousarray(_np(l["counts"])).view(np.uint32)[: l["n"]].copy()) for l in lists}
        words, poss, self.meta = [], [], []
        for i in sorted(self.lists):
            km, cn = self.lists[i]
            n = len(cn)
            if n == 0:
                continue
            a = (i + t) % self.nb
            pos = np.stack([murmur64_packed(km, self.k, self.seeds[(a * self.nh + j) % 128]) % np.uint64(self.Lbits) for j in range(self.nh)], axis=1).astype(np.int64)
            want = ((self.bin_of_occ[cn].astype(np.int64)[:, None] >> np.arange(self.nh)[None, :]) & 1)
            x = np.repeat(np.arange(n, dtype=np.int64), self.nh)
            j = np.tile(np.arange(self.nh, dtype=np.int64), n)
            words.append(_word(pos.reshape(-1), want.reshape(-1), i, x, j))
            poss.append(pos.reshape(-1))
            self.meta.append((i, n, pos, want))
            self.attempts += n
        # the commits of the round before travel in front of this round's triples (stable sort by destination keeps them there)
        self.n_commits = len(self.pending_w)
        w = np.concatenate([self.pending_w] + words)
        p = np.concatenate([self.pending_p] + poss)
        self.pending_w, self.pending_p = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        sent, self.order, counts = self._by_dest(w, p)
        dest_c = self._owner(p[: self.n_commits]) if self.n_commits else np.zeros(0, dtype=np.int64)
        commits = np.bincount(dest_c, minlength=self.world).astype(np.int64).tolist()       # (in front of each region: the sort is stable)
        return torch.from_numpy(sent.copy()), counts, commits

    def range_verdict(self, t, words, totals, commits):
        allw = _np(words)
        is_commit = allw < 0                                        # bit 63
        off = 0
        for tot, nc in zip(totals, commits):                        # the headers say the same: commits[s] commit words in front of region s
            assert is_commit[off: off + nc].all() and not is_commit[off + nc: off + tot].any()
            off += tot
        assert off == len(allw)
        self.range_commit(torch.from_numpy(np.ascontiguousarray(allw[is_commit])))       # the previous round's winners first
        ver_all = np.zeros(len(allw), dtype=np.uint8)
        tr = allw[~is_commit]
        ver = np.zeros(len(tr), dtype=np.uint8)
        if len(tr):
            pos, want, i = tr & M36, (tr >> 36) & 1, (tr >> 37) & 15
            tg, vl = np.zeros(len(tr), dtype=np.int64), np.zeros(len(tr), dtype=np.int64)
            for li in np.unique(i):
                msk = i == li
                tag, val = self._bits(int((li + t) % self.nb))
                tg[msk], vl[msk] = self._get(tag, pos[msk]), self._get(val, pos[msk])
            conflict = (tg == 1) & (vl != want)
            untag = tg == 0
            both = np.zeros(len(tr), dtype=bool)
            u = np.flatnonzero(untag)
            if len(u):
                key = (i[u] << 36) | pos[u]
                ku, inv = np.unique(key, return_inverse=True)
                has0 = np.bincount(inv, weights=(want[u] == 0), minlength=len(ku)) > 0
                has1 = np.bincount(inv, weights=(want[u] == 1), minlength=len(ku)) > 0
                both[u] = (has0 & has1)[inv]
            ver[:] = conflict.astype(np.uint8) | (untag.astype(np.uint8) << 1) | (both.astype(np.uint8) << 2)
        ver_all[~is_commit] = ver
        return torch.from_numpy(ver_all)

    def range_resolve(self, t, verdicts):
        back = _np(verdicts)
        if back.ndim == 2:                                          # fixed-size messages: capx bytes per destination, the first sent[q] of them answered
            if self.overflowed:
                back = np.zeros(len(self.order), dtype=np.uint8)    # (void build: any verdicts do)
            else:
                back = np.concatenate([back[q, : self._sent[q]] for q in range(self.world)])
        verdict = np.empty(len(back), dtype=np.uint8)
        verdict[self.order] = back                                  # the order the words were generated in: commits, then the triples
        verdict = verdict[self.n_commits:]
        cw, cp, off = [], [], 0
        for i, n, pos, want in self.meta:
            v = verdict[off: off + n * self.nh].reshape(n, self.nh)
            off += n * self.nh
            failed = (v & 1).any(axis=1)
            untag = ((v >> 1) & 1).astype(bool)
            contended = ~failed & (((v >> 2) & 1).astype(bool) & untag).any(axis=1)
            decided = {}
            for x in np.flatnonzero(contended):                     # the sequential greedy among the contended (kmodel.hpp:543-555)
                ps, ws, us = pos[x], want[x], untag[x]
                if any(us[j] and ps[j] in decided and decided[ps[j]] != ws[j] for j in range(self.nh)):
                    failed[x] = True
                    continue
                for j in range(self.nh):
                    if us[j]:
                        decided[ps[j]] = decided.get(ps[j], 0) | int(ws[j])   # a k-mer that hits a position twice leaves the OR (kmodel.hpp:611-618)
            self.n_contended += int(contended.sum())
            win = ~failed
            self.successes += int(win.sum())
            wi = np.flatnonzero(win)
            if len(wi):                                             # kmodel.hpp:548-550: the (k-2)-mer of a success goes into km_back (order-free)
                so = self.o.stats()
                kb = np.ctypeslib.as_array(O.C.cast(self.L.kmo_km_back(self.o.h), O.C.POINTER(O.C.c_uint8)), shape=(max(int(so.byte_km_back), 1),))
                if so.byte_km_back:
                    sub = (self.lists[i][0][wi] >> np.uint64(2)) & np.uint64((1 << (2 * (self.k - 2))) - 1)
                    for jj in range(self.nh - 2):
                        bp = (murmur64_packed(sub, self.k - 2, self.seeds[jj]) % np.uint64(int(so.byte_km_back) * 8)).astype(np.int64)
                        np.bitwise_or.at(kb, bp >> 3, (np.uint8(0x80) >> (bp & 7).astype(np.uint8)))
                m = untag[wi]
                p = pos[wi][m]
                # the value a winner leaves at a position: the OR over its hashes that hit it
                val = want[wi][m]
                if len(p):
                    rows = np.repeat(wi, m.sum(axis=1))
                    key = rows.astype(np.int64) * np.int64(1 << 40) + p
                    ku, inv = np.unique(key, return_inverse=True)
                    val = (np.bincount(inv, weights=val, minlength=len(ku)) > 0)[inv].astype(np.int64)
                a = (i + t) % self.nb                               # a commit word names its ARRAY and carries bit 63
                cw.append(_word(p, val, a, np.zeros(len(p), dtype=np.int64), np.zeros(len(p), dtype=np.int64)) | np.int64(-(1 << 63)))
                cp.append(p)
            km, cn = self.lists[i]
            keep = reorder(np.arange(n, dtype=np.int64), failed)
            self.lists[i] = (km[keep], cn[keep])
            if t == self.nb - 1:                                    # kmodel.hpp:567-571 + what slot 0 keeps for the final block
                km2, cn2 = self.lists[i]
                self.rest_k.append(km2.copy())
                self.rest_c.append(cn2.copy())
                self.stale[i] = (km2[:1].copy(), int(cn2[0])) if len(cn2) else None
        self.pending_w = np.concatenate(cw) if cw else np.zeros(0, dtype=np.int64)
        self.pending_p = np.concatenate(cp) if cp else np.zeros(0, dtype=np.int64)

    def range_flush(self):
        sent, _, counts = self._by_dest(self.pending_w, self.pending_p)
        self.pending_w, self.pending_p = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        return torch.from_numpy(sent.copy()), counts

    def range_commit(self, commits):
        tr = _np(commits)
        if not len(tr):
            return
        pos, v, i = tr & M36, (tr >> 36) & 1, (tr >> 37) & 15
        for li in np.unique(i):
            msk = i == li
            tag, val = self._bits(int(li))
            p = pos[msk]
            np.bitwise_or.at(tag, p >> 3, (np.uint8(0x80) >> (p & 7).astype(np.uint8)))
            pv = p[v[msk] == 1]
            np.bitwise_or.at(val, pv >> 3, (np.uint8(0x80) >> (pv & 7).astype(np.uint8)))

    # ---- fixed-size messages (kmx_range_inband): a region = [2 header words: commits | triples << 32, 0] + capx words; words beyond
    # capx are dropped and the build is void (local() reports it) -- the product's rule, so that the orchestration's fallback
    # to counted messages runs here too (RANGE_ENGINE_CAPX forces a small capacity)
    def range_inband(self):
        held = -(-self.nb // self.world)
        mean0 = held * BUCKET * self.nh // self.world
        self.capx = int(os.environ.get("RANGE_ENGINE_CAPX", mean0 + mean0 // 4 + 8192))
        self.overflowed = False

    def _pack(self, words, counts, commits):
        w = _np(words)
        msg = np.zeros((self.world, 2 + self.capx), dtype=np.int64)
        off = 0
        self._sent = list(counts)
        for q, (tot, nc) in enumerate(zip(counts, commits)):
            msg[q, 0] = nc | ((tot - nc) << 32)
            keep = min(tot, self.capx)
            if tot > self.capx:
                self.overflowed = True
            msg[q, 2: 2 + keep] = w[off: off + keep]
            off += tot
        return torch.from_numpy(msg)

    def _unpack(self, recv):
        r = _np(recv).reshape(self.world, 2 + self.capx)
        totals, commits, parts = [], [], []
        for s in range(self.world):
            nc, nt = int(r[s, 0] & 0xFFFFFFFF), int((r[s, 0] >> 32) & 0xFFFFFFFF)
            nc = min(nc, self.capx)
            nt = min(nt, self.capx - nc)
            totals.append(nc + nt); commits.append(nc); parts.append(r[s, 2: 2 + nc + nt])
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64), totals, commits

    def range_emit_inband(self, t, lists):
        words, counts, commits = self.range_emit(t, lists)
        return self._pack(words, counts, commits)

    def range_verdict_inband(self, t, recv):
        words, totals, commits = self._unpack(recv)
        if (words < 0).sum() != sum(commits):                       # a sender dropped words: nothing consistent can be answered (the build is void anyway)
            return torch.zeros((self.world, self.capx), dtype=torch.uint8)
        ver = _np(self.range_verdict(t, torch.from_numpy(words), totals, commits))
        out = np.zeros((self.world, self.capx), dtype=np.uint8)
        off = 0
        for s, tot in enumerate(totals):
            out[s, :tot] = ver[off: off + tot]
            off += tot
        return torch.from_numpy(out)

    def range_flush_inband(self):
        words, counts = self.range_flush()
        return self._pack(words, counts, counts)

    def range_commit_inband(self, recv):
        words, _, _ = self._unpack(recv)
        self.range_commit(torch.from_numpy(words[words < 0]))

    def local(self):
        st = _Local()
        st.reserved = int(getattr(self, "overflowed", False))
        st.attempts, st.successes = self.attempts + self.extra_attempts, self.successes
        st.fast_commits, st.contended, st.finisher_iters = self.successes, self.n_contended, 0
        km = np.concatenate(self.rest_k) if self.rest_k else np.zeros(0, dtype=np.uint64)
        cn = np.concatenate(self.rest_c) if self.rest_c else np.zeros(0, dtype=np.uint32)
        st.rest_entries = len(cn)
        return st, torch.from_numpy(km.view(np.int64)), torch.from_numpy(cn.view(np.int32))

    def array_ranges(self, a):
        so = self.o.stats()
        nbytes = int(so.km_byte_size)
        bounds = [min(2 * c, nbytes) for c in self.cell_lo]
        bounds[-1] = nbytes
        return [(v, bounds) for v in self.array_views(a)]
