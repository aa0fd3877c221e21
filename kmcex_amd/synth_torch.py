"""Device-side twin of synth.py (k <= 31, one word per k-mer) for benchmark-size streams.

torch has no uint64 arithmetic, so everything runs on int64 bit patterns: wrap-around multiply/add are the
same bits, logical right shifts are an arithmetic shift + mask.  tests/test_synth.py pins this to synth.py.
"""
from __future__ import annotations

import torch


def _s(x: int) -> int:
    """uint64 constant -> the int64 with the same bits."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(z: torch.Tensor, s: int) -> torch.Tensor:
    return (z >> s) & ((1 << (64 - s)) - 1)


def splitmix64(x: torch.Tensor) -> torch.Tensor:
    z = x + _s(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _s(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _umod(u: torch.Tensor, m: int) -> torch.Tensor:
    """unsigned 64-bit u mod m on int64 bit patterns"""
    return ((_lsr(u, 1) % m) * 2 + (u & 1)) % m


def _rev2(x: torch.Tensor) -> torch.Tensor:
    for mask, sh in ((0x3333333333333333, 2), (0x0F0F0F0F0F0F0F0F, 4), (0x00FF00FF00FF00FF, 8), (0x0000FFFF0000FFFF, 16)):
        x = (_lsr(x, sh) & mask) | ((x & mask) << sh)
    return _lsr(x, 32) | (x << 32)


def revcomp(km: torch.Tensor, k: int) -> torch.Tensor:
    assert k <= 31
    return _lsr(_rev2(~km), 64 - 2 * k)


def random_kmers(n: int, k: int, seed_k: int, device, start: int = 0) -> torch.Tensor:
    assert k <= 31
    i = torch.arange(start, start + n, dtype=torch.int64, device=device)
    return splitmix64(i + seed_k) & ((1 << (2 * k)) - 1)


def d1_counts_range(start: int, n: int, ci: int, cs: int, seed_c: int, device) -> torch.Tensor:
    """D1 counts of the ranks start .. start+n-1"""
    u = splitmix64(torch.arange(start, start + n, dtype=torch.int64, device=device) + seed_c)
    sel = _umod(u, 10)
    v = _lsr(u, 8)
    a = v % 4
    b = v % max(60 - ci + 1, 1)
    c = v % (cs - ci + 1)
    out = torch.where(sel < 5, a, torch.where(sel < 9, b, c)) + ci
    return torch.clamp(out, max=cs).to(torch.int32)


def d1_counts(n: int, ci: int, cs: int, seed_c: int, device) -> torch.Tensor:
    return d1_counts_range(0, n, ci, cs, seed_c, device)


def make_stream(n: int, k: int, ci: int, cs: int, device, seed_k: int = 1, seed_c: int = 2, range_values: int = 1 << 30, chunk: int = 1 << 28):
    """Sorted distinct canonical k-mers (int64 holding the uint64 bits; k <= 31 keeps them non-negative)
    and D1 counts (int32 holding uint32 bits).

    torch.unique handles fewer than 2^31 elements, so a larger stream is produced in value ranges: every range
    regenerates the draws chunk by chunk (cheap), keeps the canonical values that fall into it, and sorts them; the
    ranges are concatenated in order.  Same stream as the one-shot path (tests/test_synth.py)."""
    parts = max(1, -(-n // range_values))                    # ~2^30 values per range
    if parts == 1:
        x = random_kmers(n, k, seed_k, device)
        x = torch.minimum(x, revcomp(x, k))
        x = torch.unique(x, sorted=True)
        return x, d1_counts(x.numel(), ci, cs, seed_c, device)
    span = 1 << (2 * k)                                      # canonical values are skewed to the low half, so cut by quantiles
    probe = random_kmers(min(n, 1 << 22), k, seed_k, device)
    probe = torch.minimum(probe, revcomp(probe, k)).sort().values
    cuts = [0] + [int(probe[(len(probe) * p) // parts]) for p in range(1, parts)] + [span]
    del probe
    out = []
    for p in range(parts):
        keep = []
        for start in range(0, n, chunk):
            x = random_kmers(min(chunk, n - start), k, seed_k, device, start=start)
            x = torch.minimum(x, revcomp(x, k))
            keep.append(x[(x >= cuts[p]) & (x < cuts[p + 1])])
        out.append(torch.unique(torch.cat(keep), sorted=True))
        del keep
    x = torch.cat(out)
    del out
    cnt = torch.cat([d1_counts_range(s0, min(chunk, x.numel() - s0), ci, cs, seed_c, device) for s0 in range(0, x.numel(), chunk)])
    return x, cnt
