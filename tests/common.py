"""Shared by the golden generator and the tests: the case table and the query-set recipe."""
import hashlib

import numpy as np

from kmcex_amd import synth

# name, k, ci, cs, nh, nb, n_draws
CASES = [
    ("tiny_k31", 31, 1, 1023, 7, 5, 20000),
    ("k31_ci2_200k", 31, 2, 1023, 7, 5, 200000),
    ("k55_nh9_nb6", 55, 1, 4095, 9, 6, 30000),
    ("k21_nh6_nb3", 21, 2, 255, 6, 3, 5000),
    ("k31_multiblock_ci1", 31, 1, 1023, 7, 5, 3200000),      # 2 full blocks + partial with unused rows (Q1)
    ("k31_multiblock_ci2", 31, 2, 1023, 7, 5, 4000000),      # RS-scale plumbing stand-in (BASELINE configs[0])
    ("k55_multiblock", 55, 1, 4095, 9, 6, 2000000),          # 1 full block (6*2^18) + partial, two-word k-mers
    ("k32_nb4", 32, 1, 1023, 7, 4, 50000),                   # k == 32 boundary, pre_len 4
    ("k33_first_two_word", 33, 1, 1023, 7, 5, 30000),        # first k that needs two words (k > 32 canonicalisation quirk Q4)
    ("k64_cs65535_nh8_nb3", 64, 2, 65535, 8, 3, 30000),      # largest k, 2-byte KMC counters, three Bloom classes
    ("k16_nh5_nb2", 16, 1, 255, 5, 2, 20000),                # short k-mers, few hashes, two arrays
    ("k27_ci3_nb8", 27, 3, 1023, 7, 8, 40000),               # ci = 3: three Bloom classes start at count 3; eight arrays
    ("k31_nh3_nb1", 31, 1, 255, 3, 1, 20000),                # smallest nh, a single array
    ("k31_nh12_cs8191", 31, 1, 8191, 12, 2, 20000),          # twelve hashes (4096 bins), the wide-template kernels at k <= 32
]
# KMC2-layout databases (what KMC 3 emits): bin-major listing, not globally sorted -> the insert order differs
# name, k, ci, cs, nh, nb, n_draws, n_bins
KMC2_CASES = [
    ("k31_kmc2_6bins", 31, 1, 1023, 7, 5, 1600000, 6),
    ("k55_kmc2_3bins", 55, 2, 4095, 9, 6, 40000, 3),
]
# genome-like streams (overlapping k-mers of a random sequence): name, k, ci, cs, nh, nb, n_bases
GENOME_CASES = [
    ("genome_k31_ci1", 31, 1, 1023, 7, 5, 400000),
    ("genome_k27_ci2", 27, 2, 1023, 7, 4, 250000),
]
CASE = {c[0]: c for c in CASES}
SMALL = ["tiny_k31", "k31_ci2_200k", "k55_nh9_nb6", "k21_nh6_nb3", "k32_nb4", "k33_first_two_word", "k64_cs65535_nh8_nb3", "k16_nh5_nb2", "k27_ci3_nb8", "k31_nh3_nb1", "k31_nh12_cs8191"]
LARGE = ["k31_multiblock_ci1", "k31_multiblock_ci2", "k55_multiblock"]
MAX_PRESENT = 400000
ABSENT_SEED = 0xABCDEF0123                     # far outside the index range of any stream, so these draws are absent


def query_set(km, k, seed=7, max_present=MAX_PRESENT):
    """Inserted k-mers in a seeded shuffle, first half reverse-complemented, + 10 % absent draws."""
    rng = np.random.default_rng(seed)
    idx = rng.permutation(len(km))
    if max_present is not None:
        idx = idx[:max_present]
    q = km[idx].copy()
    h = len(q) // 2
    q[:h] = synth.revcomp(q[:h], k)
    absent = synth.random_kmers(max(len(q) // 10, 10), k, seed_k=ABSENT_SEED)
    return np.concatenate([q, absent])


def genome_query_set(km, k):
    """every stored k-mer (half reverse-complemented) + all 8 de Bruijn neighbours of a sample (mostly absent)"""
    q = km.copy()
    q[::2] = synth.revcomp(q[::2], k)
    s = km[::5]
    mask = np.uint64((1 << (2 * k)) - 1) if k < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    nb = [((s << np.uint64(2)) | np.uint64(x)) & mask for x in range(4)] + [(s >> np.uint64(2)) | (np.uint64(x) << np.uint64(2 * (k - 1))) for x in range(4)]
    return np.concatenate([q] + nb)


def sha_file(p):
    h = hashlib.sha256()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def sha_occ(occ):
    return hashlib.sha256(np.asarray(occ).astype("<i4").tobytes()).hexdigest()
