"""Deterministic synthetic k-mer streams -- host/numpy plumbing.

The reference has no data generator; its inputs are KMC databases produced from FASTQ by an
external binary that is absent here (main.cpp:137-140).  Benchmarks and tests therefore use a
seeded synthetic stream that has the *shape* of a KMC1 listing: distinct canonical k-mers in
ascending 2-bit order (kmc_file.cpp:428-515 lists prefix-major, suffix-ascending) with counts
drawn from the "D1" mixture.

Packed k-mer layout used everywhere in this repo (host and device): ``W = ceil(k/32)`` uint64
words per k-mer, word 0 most significant, holding the 2k-bit integer right-aligned with
A=0,C=1,G=2,T=3 and the first base in the most significant position (tools.hpp:63-76).
Arrays are ``uint64[n, W]`` (or ``uint64[n]`` when W == 1).
"""
from __future__ import annotations

import numpy as np

MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser applied to ``x + golden`` (vectorised, wrap-around uint64)."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def words_for_k(k: int) -> int:
    return (k + 31) // 32


def _rev2_u64(x: np.ndarray) -> np.ndarray:
    """Reverse the order of the 32 two-bit groups of each uint64."""
    x = x.astype(np.uint64)
    m = [(0x3333333333333333, 2), (0x0F0F0F0F0F0F0F0F, 4), (0x00FF00FF00FF00FF, 8),
         (0x0000FFFF0000FFFF, 16)]
    for mask, sh in m:
        mask = np.uint64(mask)
        sh = np.uint64(sh)
        x = ((x >> sh) & mask) | ((x & mask) << sh)
    return (x >> np.uint64(32)) | (x << np.uint64(32))


def revcomp(kmers: np.ndarray, k: int) -> np.ndarray:
    """True reverse complement of packed k-mers (any k <= 64)."""
    W = words_for_k(k)
    a = kmers.reshape(-1, W)
    if W == 1:
        r = _rev2_u64(~a[:, 0]) >> np.uint64(64 - 2 * k)
        return r.reshape(kmers.shape)
    assert W == 2
    hi, lo = a[:, 0], a[:, 1]
    # 128-bit value v = hi:lo ; reverse groups -> (rev(lo) : rev(hi)), complement, shift right by 128-2k
    nh, nl = _rev2_u64(~lo), _rev2_u64(~hi)
    s = 128 - 2 * k
    assert 0 <= s < 64
    out = np.empty_like(a)
    if s == 0:
        out[:, 1], out[:, 0] = nl, nh
    else:
        s_, c_ = np.uint64(s), np.uint64(64 - s)
        out[:, 1] = (nl >> s_) | (nh << c_)
        out[:, 0] = nh >> s_
    return out.reshape(kmers.shape)


def _less(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Lexicographic a < b on (n, W) word arrays."""
    if a.ndim == 1:
        return a < b
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for w in range(a.shape[1]):
        lt |= eq & (a[:, w] < b[:, w])
        eq &= a[:, w] == b[:, w]
    return lt


def canonical(kmers: np.ndarray, k: int) -> np.ndarray:
    rc = revcomp(kmers, k)
    take_rc = _less(rc, kmers)
    out = kmers.copy()
    out[take_rc] = rc[take_rc]
    return out


def sort_unique(kmers: np.ndarray) -> np.ndarray:
    if kmers.ndim == 1:
        return np.unique(kmers)
    order = np.lexsort([kmers[:, w] for w in range(kmers.shape[1] - 1, -1, -1)])
    s = kmers[order]
    keep = np.ones(len(s), dtype=bool)
    keep[1:] = np.any(s[1:] != s[:-1], axis=1)
    return s[keep]


def random_kmers(n: int, k: int, seed_k: int = 1, start: int = 0) -> np.ndarray:
    """``n`` raw draws (not canonical, not unique): x_i = splitmix64(seed_k + i) & (4^k - 1)."""
    W = words_for_k(k)
    i = np.arange(start, start + n, dtype=np.uint64)
    if W == 1:
        x = splitmix64(np.uint64(seed_k) + i)
        if k < 32:
            x &= np.uint64((1 << (2 * k)) - 1)
        return x
    out = np.empty((n, 2), dtype=np.uint64)
    out[:, 1] = splitmix64(np.uint64(seed_k) + np.uint64(2) * i)
    out[:, 0] = splitmix64(np.uint64(seed_k) + np.uint64(2) * i + np.uint64(1)) & np.uint64((1 << (2 * k - 64)) - 1)
    return out


def d1_counts(n: int, ci: int, cs: int, seed_c: int = 2) -> np.ndarray:
    """The "D1" count mixture: 50 % uniform [ci, ci+3], 40 % uniform [ci, 60], 10 % uniform [ci, cs]."""
    u = splitmix64(np.uint64(seed_c) + np.arange(n, dtype=np.uint64))
    sel = (u % np.uint64(10)).astype(np.int64)
    v = (u >> np.uint64(8))
    a = (v % np.uint64(4)).astype(np.int64)
    b = (v % np.uint64(max(60 - ci + 1, 1))).astype(np.int64)
    c = (v % np.uint64(cs - ci + 1)).astype(np.int64)
    out = np.where(sel < 5, a, np.where(sel < 9, b, c)) + ci
    return np.minimum(out, cs).astype(np.uint32)


def make_stream(n: int, k: int, ci: int, cs: int, seed_k: int = 1, seed_c: int = 2):
    """Sorted distinct canonical k-mers (KMC1 listing order) + D1 counts by rank."""
    km = sort_unique(canonical(random_kmers(n, k, seed_k), k))
    return km, d1_counts(len(km), ci, cs, seed_c)


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def to_ascii(kmers: np.ndarray, k: int) -> np.ndarray:
    """Packed -> uint8[n, k] ASCII (first base first)."""
    W = words_for_k(k)
    a = kmers.reshape(-1, W)
    n = a.shape[0]
    out = np.empty((n, k), dtype=np.uint8)
    for pos in range(k):
        bit = 2 * (k - 1 - pos)            # bit offset from the LSB of the 2k-bit integer
        w = W - 1 - bit // 64
        out[:, pos] = _ACGT[((a[:, w] >> np.uint64(bit % 64)) & np.uint64(3)).astype(np.int64)]
    return out


def to_strings(kmers: np.ndarray, k: int):
    return [r.tobytes().decode() for r in to_ascii(kmers, k)]


def from_strings(strs, k: int) -> np.ndarray:
    """ASCII (ACGT only) -> packed."""
    W = words_for_k(k)
    arr = np.frombuffer("".join(strs).encode(), dtype=np.uint8).reshape(-1, k)
    code = np.zeros(256, dtype=np.uint64)
    code[ord("C")] = 1
    code[ord("G")] = 2
    code[ord("T")] = 3
    out = np.zeros((arr.shape[0], W), dtype=np.uint64)
    for pos in range(k):
        bit = 2 * (k - 1 - pos)
        w = W - 1 - bit // 64
        out[:, w] |= code[arr[:, pos]] << np.uint64(bit % 64)
    return out[:, 0] if W == 1 else out


def genome_stream(n_bases: int, k: int, ci: int, cs: int, seed: int = 11, seed_c: int = 2):
    """Genome-like stream: all overlapping k-mers of a random sequence, canonicalised, distinct, in listing order.

    Unlike independent draws these k-mers have their de Bruijn neighbours in the set, which is what the query's
    neighbour-based disambiguation (kmodel.hpp:286-359) feeds on."""
    assert k <= 32
    rng = np.random.default_rng(seed)
    bases = rng.integers(0, 4, size=n_bases, dtype=np.uint64)
    n = n_bases - k + 1
    v = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        v = (v << np.uint64(2)) | bases[j:j + n]
    if k < 32:
        v &= np.uint64((1 << (2 * k)) - 1)
    km = sort_unique(canonical(v, k))
    return km, d1_counts(len(km), ci, cs, seed_c)
