"""Rehearsal of the NORTH STAR's multi-GPU partition: every coupled array cut by POSITION RANGE across
the ranks, k-mers routed to their owning shards by all-to-all -- run for real over torch.distributed (gloo, CPU) with plain
numpy as the per-rank engine, checked bit for bit against the CPU oracle, and COUNTED: bytes on the wire and collectives
per build.  This is measurement / test infrastructure (tools/, tests/): the product's multi-GPU path is the ring of whole
arrays (kmcex_amd/dist.py); DESIGN.md §5 quotes the numbers this prints to say why.

The protocol (exact: it reproduces the sequential greedy of kmodel.hpp:543-622 for any number of ranks):
  * rank r owns positions [r*L/P, (r+1)*L/P) of EVERY array (tag + value bits); list i of a block lives on rank i % P, which
    hashes its k-mers, keeps the list order and does the reorder (kmodel.hpp:529-540) locally -- k-mers never move.
  * a round (all nb lists at once, list i against array (i + t) % nb) is three all-to-alls:
      1. triples (array, position, wanted value, list slot) -> range owners;
      2. verdict per triple back: conflict with a set tag | position untagged | untagged and wanted with BOTH values by
         triples of this round (the owner sees every claim on its positions, so contention is found where the bits live);
      3. commits (array, position, value) of the winners -> range owners.
    A candidate none of whose untagged positions is contended wins outright; the contended ones (a superset of the truly
    order-dependent k-mers: failed claimants count too) are decided in list order on the list's rank from the verdicts alone
    -- the only writers that can matter to them are earlier contended winners.
usage: python tools/range_shard_rehearsal.py <case of tests/common.py CASES> <world>     (spawns the ranks itself)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BUCKET = 1 << 18
M64 = np.uint64(0xC6A4A7935BD1E995)
R47 = np.uint64(47)


def murmur64_packed(km, k, seed):
    """MurmurHash64A (tools.hpp:16-50) of the ASCII strings of packed k-mers (k <= 32), vectorised; seed: uint32"""
    km = km.astype(np.uint64)
    n = len(km)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    chars = np.empty((n, k), dtype=np.uint8)
    for i in range(k):
        chars[:, i] = lut[((km >> np.uint64(2 * (k - 1 - i))) & np.uint64(3)).astype(np.int64)]
    with np.errstate(over="ignore"):
        h = np.full(n, np.uint64(seed) ^ (np.uint64(k) * M64), dtype=np.uint64)
        nblk = k // 8
        for b in range(nblk):
            w = np.zeros(n, dtype=np.uint64)
            for j in range(8):
                w |= chars[:, 8 * b + j].astype(np.uint64) << np.uint64(8 * j)
            w *= M64
            w ^= w >> R47
            w *= M64
            h ^= w
            h *= M64
        rem = k & 7
        if rem:
            for j in range(rem):
                h ^= chars[:, 8 * nblk + j].astype(np.uint64) << np.uint64(8 * j)
            h *= M64
        h ^= h >> R47
        h *= M64
        h ^= h >> R47
    return h


def reorder(entries, failed):
    """reorder_buffer (kmodel.hpp:529-540): survivors below m stay; the i-th hole from the left gets the i-th survivor from the right"""
    n = len(entries)
    m = int(failed.sum())
    out = entries[:m].copy()
    holes = np.flatnonzero(~failed[:m])
    movers = np.flatnonzero(failed[m:])[::-1] + m
    out[holes] = entries[movers[: len(holes)]]
    return out


class Wire:
    """all_to_all_v of int64 rows over torch.distributed, counting what leaves the rank"""

    def __init__(self):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        self.bytes_sent = 0
        self.collectives = 0

    def exchange(self, rows, dest):
        """rows [m, c] int64, dest [m] -> rows received (grouped by source rank, order kept), and the permutation to answer"""
        import torch
        order = np.argsort(dest, kind="stable")
        send = np.ascontiguousarray(rows[order])
        splits = np.bincount(dest, minlength=self.world).astype(np.int64)
        t_splits = torch.from_numpy(splits.copy())
        r_splits = torch.empty_like(t_splits)
        self.dist.all_to_all_single(r_splits, t_splits)
        c = rows.shape[1]
        out = torch.empty((int(r_splits.sum()), c), dtype=torch.int64)
        self.dist.all_to_all_single(out, torch.from_numpy(send).reshape(-1, c), r_splits.tolist(), splits.tolist())
        self.collectives += 1                                           # (+ the split sizes: a fixed-size exchange, not counted as data)
        self.bytes_sent += int((splits.sum() - splits[self.rank]) * c * 8)
        return out.numpy(), order, splits, r_splits.numpy()

    def answer(self, rows, r_splits, splits):
        """send one row per received row back to where it came from; returns them in the order of the original send"""
        import torch
        c = rows.shape[1]
        out = torch.empty((int(splits.sum()), c), dtype=torch.int64)
        self.dist.all_to_all_single(out, torch.from_numpy(np.ascontiguousarray(rows)).reshape(-1, c), splits.tolist(), r_splits.tolist())
        self.collectives += 1
        self.bytes_sent += int((r_splits.sum() - r_splits[self.rank]) * c * 8)
        return out.numpy()


def build_range_sharded(wire, k, nh, nb, L, seeds, kmers, bins):
    """kmers / bins: the coupled-array stream in listing order (every rank passes the same arrays; a rank only works on the
    lists it holds).  Returns (tag bits, value bits of this rank's range per array, survivors (stream indices), attempts, successes)."""
    P, r = wire.world, wire.rank
    lo, hi = r * L // P, (r + 1) * L // P
    tag = [np.zeros(hi - lo, dtype=np.uint8) for _ in range(nb)]
    val = [np.zeros(hi - lo, dtype=np.uint8) for _ in range(nb)]
    bounds = np.array([(q + 1) * L // P for q in range(P)], dtype=np.int64)

    def owner_of(pos):
        return np.searchsorted(bounds, pos, side="right").astype(np.int64)

    n_km = len(kmers)
    blk = nb * BUCKET
    survivors, attempts, successes = [], 0, 0
    for b0 in range(0, n_km, blk):
        lists = {}
        for i in range(nb):
            s, e = b0 + i * BUCKET, min(b0 + (i + 1) * BUCKET, n_km)
            if s < e and i % P == r:
                lists[i] = np.arange(s, e, dtype=np.int64)          # entries = indices into the stream, in list order
        for t in range(nb):
            # 1. triples of every list this rank holds
            rows, meta = [], []
            for i, ent in lists.items():
                if len(ent) == 0:
                    continue
                a = (i + t) % nb
                km = kmers[ent]
                pos = np.stack([murmur64_packed(km, k, seeds[(a * nh + j) % 128]) % np.uint64(L) for j in range(nh)], axis=1).astype(np.int64)
                want = ((bins[ent][:, None] >> np.arange(nh)[None, :]) & 1).astype(np.int64)
                n = len(ent)
                slot = np.repeat(np.arange(n, dtype=np.int64), nh)
                rows.append(np.stack([np.full(n * nh, a, dtype=np.int64), pos.reshape(-1), want.reshape(-1), (np.int64(i) << 32) | slot], axis=1))
                meta.append((i, n, pos, want))
                attempts += n
            rows = np.concatenate(rows) if rows else np.zeros((0, 4), dtype=np.int64)
            got, order, splits, r_splits = wire.exchange(rows, owner_of(rows[:, 1]) if len(rows) else np.zeros(0, dtype=np.int64))
            # 2. verdicts where the bits live
            ver = np.zeros((len(got), 1), dtype=np.int64)
            if len(got):
                a_, p_, w_ = got[:, 0], got[:, 1] - lo, got[:, 2]
                tg, vl = np.zeros(len(got), dtype=np.int64), np.zeros(len(got), dtype=np.int64)
                for a in range(nb):
                    msk = a_ == a
                    tg[msk] = tag[a][p_[msk]]
                    vl[msk] = val[a][p_[msk]]
                conflict = (tg == 1) & (vl != w_)
                untagged = tg == 0
                key = a_ * np.int64(L) + got[:, 1]
                both = np.zeros(len(got), dtype=bool)
                u = np.flatnonzero(untagged)
                if len(u):
                    ku, inv = np.unique(key[u], return_inverse=True)
                    has0 = np.bincount(inv, weights=(w_[u] == 0), minlength=len(ku)) > 0
                    has1 = np.bincount(inv, weights=(w_[u] == 1), minlength=len(ku)) > 0
                    both[u] = (has0 & has1)[inv]
                ver[:, 0] = conflict.astype(np.int64) | (untagged.astype(np.int64) << 1) | (both.astype(np.int64) << 2)
            back = wire.answer(ver, r_splits, splits)
            verdict = np.empty(len(rows), dtype=np.int64)
            verdict[order] = back[:, 0]
            # 3. winners; contended candidates in list order
            commits, off = [], 0
            for i, n, pos, want in meta:
                v = verdict[off: off + n * nh].reshape(n, nh)
                off += n * nh
                a = (i + t) % nb
                failed = (v & 1).any(axis=1)
                untag = (v >> 1) & 1
                contended = ~failed & (((v >> 2) & 1) & untag).any(axis=1)
                decided = {}
                for x in np.flatnonzero(contended):                      # the sequential greedy, among the contended only
                    ps, ws, us = pos[x], want[x], untag[x]
                    mine = {}
                    for j in range(nh):
                        if us[j]:
                            mine[ps[j]] = mine.get(ps[j], 0) | int(ws[j])     # a k-mer that hits a position twice leaves the OR (kmodel.hpp:611-618)
                    # (the CHECK compares each hash's own value with what is there: kmodel.hpp:604-610)
                    if any(us[j] and ps[j] in decided and decided[ps[j]] != ws[j] for j in range(nh)):
                        failed[x] = True
                        continue
                    decided.update(mine)
                win = ~failed
                successes += int(win.sum())
                wi = np.flatnonzero(win)
                if len(wi):
                    m = untag[wi].astype(bool)
                    cp = pos[wi][m]
                    cw = want[wi][m]
                    commits.append(np.stack([np.full(len(cp), a, dtype=np.int64), cp, cw], axis=1))
                lists[i] = reorder(lists[i], failed)
            crow = np.concatenate(commits) if commits else np.zeros((0, 3), dtype=np.int64)
            got, _, _, _ = wire.exchange(crow, owner_of(crow[:, 1]) if len(crow) else np.zeros(0, dtype=np.int64))
            for a in range(nb):
                msk = got[:, 0] == a
                if msk.any():
                    p = got[msk, 1] - lo
                    tag[a][p] = 1
                    np.bitwise_or.at(val[a], p, got[msk, 2].astype(np.uint8))
        for i in sorted(lists):
            survivors.extend(lists[i].tolist())
    return tag, val, (lo, hi), survivors, attempts, successes


def worker(rank, world, port, case, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        torch.set_num_threads(1)
        import oracle_lib as O
        from common import CASE
        from kmcex_amd import synth
        dist.init_process_group("gloo", rank=rank, world_size=world)
        name, k, ci, cs, nh, nb, n_draw = CASE[case]
        km, cnt = synth.make_stream(n_draw, k, ci, cs)
        o = O.OracleModel(ci, cs, nh, nb)
        o.build(k, km, cnt)
        so = o.stats()
        bf_num = 1 if ci == 1 else 3
        sel = cnt >= ci + bf_num                                        # the coupled-array class (kmodel.hpp:70-73)
        bin_of_occ, _ = O.occubin_table(cs + 1, nh)
        L = int(so.km_byte_size) * 8
        seeds = [int(O.lib().kmo_hash_seed(i)) for i in range(128)]
        wire = Wire()
        tag, val, (lo, hi), surv, attempts, successes = build_range_sharded(wire, k, nh, nb, L, seeds, km[sel], bin_of_occ[cnt[sel]].astype(np.int64))
        # compare this rank's range of every array with the oracle's bytes (MSB-first bits, kmodel.hpp:576-588)
        ok = True
        for a in range(nb):
            ot = np.unpackbits(o.array_bytes("tag", a))[lo:hi]
            ov = np.unpackbits(o.array_bytes("value", a))[lo:hi]
            ok &= bool(np.array_equal(ot, tag[a]) and np.array_equal(ov, val[a]))
        tot = torch.tensor([attempts, successes, len(surv), wire.bytes_sent], dtype=torch.int64)
        dist.all_reduce(tot)
        q.put({"rank": rank, "ok": ok, "attempts": int(tot[0]), "successes": int(tot[1]), "survivors": int(tot[2]), "bytes_all_ranks": int(tot[3]),
               "collectives": wire.collectives, "n_km": int(sel.sum()), "oracle": (int(so.attempts), int(so.successes), int(so.rest_entries)), "rounds": -(-int(sel.sum()) // (nb * BUCKET)) * nb})
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})


def run(case, world, timeout=900):
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ps = [ctx.Process(target=worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=timeout))
            if "error" in res[-1]:
                break
    finally:
        bad = any("error" in x for x in res) or len(res) < world
        for p in ps:
            p.join(timeout=1 if bad else 60)
            if p.is_alive():
                p.kill()
    res.sort(key=lambda x: x["rank"])
    return res


if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "tiny_k31"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    res = run(case, world)
    for x in res:
        if "error" in x:
            sys.exit(x["error"])
    r0 = res[0]
    exact = all(x["ok"] for x in res) and (r0["successes"], r0["survivors"]) == (r0["oracle"][1], r0["n_km"] - r0["oracle"][1])
    print(f"{case} on {world} ranks, arrays cut by position range: {'BIT-EXACT vs the oracle' if exact else 'MISMATCH'}; "
          f"{r0['n_km']} coupled k-mers, {r0['attempts']} attempts, {r0['rounds']} rounds, {r0['collectives']} data all-to-alls "
          f"({r0['collectives'] / max(r0['rounds'], 1):.1f} per round), {r0['bytes_all_ranks']} bytes left their rank "
          f"= {r0['bytes_all_ranks'] / max(r0['attempts'], 1):.0f} B per attempt")
    sys.exit(0 if exact else 1)
