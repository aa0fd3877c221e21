"""KMC1-format database writer (numpy) -- test/bench plumbing.

Written from the format description in SURVEY.md Appendix B.3 (what the reference's vendored
reader consumes in listing mode: kmc_file.cpp:66-99 OpenForListing, :236-290 KMC1 header parse,
:428-515 ReadNextKmer).  The reference has no writer (databases come from the external KMC
binary, main.cpp:137), so this lets the real reference (oracle/_ref), the CPU oracle and the HIP
product all eat the same synthetic input.

Layout
  <db>.kmc_pre : "KMCP" | u64 LUT[4^p] | 64-byte header | u32 header_offset(=64) | "KMCP"
  <db>.kmc_suf : "KMCS" | records | "KMCS";  record = (k-p)/4 suffix bytes (4 bases/byte,
                 first base in the top 2 bits) + counter_size little-endian count bytes.
"""
from __future__ import annotations

import struct

import numpy as np

from .synth import words_for_k


def lut_prefix_len(k: int) -> int:
    """Largest p in 7..1 with (k - p) % 4 == 0 (the reader only needs divisibility by 4)."""
    for p in range(7, 0, -1):
        if (k - p) % 4 == 0:
            return p
    raise ValueError(k)


def _suffix_bytes(kmers: np.ndarray, k: int, p: int) -> np.ndarray:
    """uint8[n, (k-p)/4]: big-endian bytes of the low 2(k-p) bits."""
    W = words_for_k(k)
    a = kmers.reshape(-1, W)
    nb = (k - p) // 4
    out = np.empty((a.shape[0], nb), dtype=np.uint8)
    for j in range(nb):
        bit = 8 * (nb - 1 - j)
        w = W - 1 - bit // 64
        out[:, j] = ((a[:, w] >> np.uint64(bit % 64)) & np.uint64(0xFF)).astype(np.uint8)
    return out


def _prefix(kmers: np.ndarray, k: int, p: int) -> np.ndarray:
    W = words_for_k(k)
    a = kmers.reshape(-1, W)
    sh = 2 * (k - p)
    if W == 1:
        return (a[:, 0] >> np.uint64(sh)).astype(np.int64)
    if sh >= 64:
        return (a[:, 0] >> np.uint64(sh - 64)).astype(np.int64)
    return ((a[:, 0] << np.uint64(64 - sh)) | (a[:, 1] >> np.uint64(sh))).astype(np.int64)   # prefix straddles the two words (k = 33..35)


def write_kmc1(path_prefix: str, kmers: np.ndarray, counts: np.ndarray, k: int, ci: int, cs: int,
               counter_size: int | None = None, total_override: int | None = None) -> None:
    """Write ``<path_prefix>.kmc_pre/.kmc_suf``.  ``kmers`` must already be in listing order."""
    p = lut_prefix_len(k)
    if counter_size is None:
        counter_size = 1
        while cs >= (1 << (8 * counter_size)):
            counter_size += 1
    n = len(counts)
    suf = _suffix_bytes(kmers, k, p)
    rec = np.empty((n, suf.shape[1] + counter_size), dtype=np.uint8)
    rec[:, :suf.shape[1]] = suf
    c = counts.astype(np.uint64)
    for b in range(counter_size):
        rec[:, suf.shape[1] + b] = ((c >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
    with open(path_prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        f.write(rec.tobytes())
        f.write(b"KMCS")
    pre = _prefix(kmers, k, p)
    lut = np.searchsorted(pre, np.arange(4 ** p, dtype=np.int64), side="left").astype(np.uint64)
    total = n if total_override is None else total_override
    hdr = struct.pack("<IIIIIIQB3xI", k, 0, counter_size, p, ci, cs & 0xFFFFFFFF, total, 0, cs >> 32)
    hdr = hdr + b"\0" * (64 - len(hdr))          # zero pad; last 4 bytes = kmc_version 0 (KMC1)
    with open(path_prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        f.write(hdr)
        f.write(struct.pack("<I", 64))
        f.write(b"KMCP")


def write_kmc2(path_prefix: str, kmers: np.ndarray, counts: np.ndarray, k: int, ci: int, cs: int, n_bins: int = 4,
               signature_len: int = 5, counter_size: int | None = None):
    """Write a KMC2-layout database (version 0x200, what KMC 3 itself emits; kmc_file.cpp:188-235) and return the
    listing order as an index array into the input.

    KMC2 keeps one LUT per bin: records are stored bin-major and sorted only inside a bin, so the listing -- and
    therefore the insert order -- is NOT globally sorted.  The bin of a k-mer is decided by its minimiser signature in
    KMC; in listing mode the reader never looks at signatures, so any assignment gives a valid database.  Here:
    bin = splitmix-ish hash of the k-mer modulo n_bins.
    """
    p = lut_prefix_len(k)
    if counter_size is None:
        counter_size = 1
        while cs >= (1 << (8 * counter_size)):
            counter_size += 1
    W = words_for_k(k)
    a = kmers.reshape(-1, W)
    n = len(counts)
    with np.errstate(over="ignore"):
        h = (a[:, W - 1] * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)
    bins = (h % np.uint64(n_bins)).astype(np.int64)
    order = np.argsort(bins, kind="stable")                     # input is sorted, so each bin stays sorted
    km_o, cnt_o, bins_o = a[order], counts[order], bins[order]
    suf = _suffix_bytes(km_o if W > 1 else km_o[:, 0], k, p)
    rec = np.empty((n, suf.shape[1] + counter_size), dtype=np.uint8)
    rec[:, :suf.shape[1]] = suf
    c = cnt_o.astype(np.uint64)
    for b in range(counter_size):
        rec[:, suf.shape[1] + b] = ((c >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
    with open(path_prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        f.write(rec.tobytes())
        f.write(b"KMCS")
    pre = _prefix(km_o if W > 1 else km_o[:, 0], k, p)
    key = bins_o * (4 ** p) + pre                                # (bin, prefix) ascending along the file
    lut = np.searchsorted(key, np.arange(n_bins * 4 ** p, dtype=np.int64), side="left").astype(np.uint64)
    sig_map = np.zeros(4 ** signature_len + 1, dtype=np.uint32)  # signature -> bin (unused by the listing reader)
    hdr = struct.pack("<IIIIIIIQB", k, 0, counter_size, p, signature_len, ci, cs & 0xFFFFFFFF, n, 0)
    hdr = hdr + b"\0" * (60 - len(hdr)) + struct.pack("<I", 0x200)   # version sits in the last 4 header bytes
    with open(path_prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        f.write(struct.pack("<Q", n))                            # guard entry after the LUTs
        f.write(sig_map.tobytes())
        f.write(hdr)
        f.write(struct.pack("<I", 64))
        f.write(b"KMCP")
    return order
