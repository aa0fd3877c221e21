"""CPU: the oracle (oracle/kmx_oracle.c) against the committed goldens made by the REAL reference
(tests/golden/make_golden.py), plus the known-answer vectors of SURVEY.md Appendix C."""
import os

import numpy as np
import pytest

import oracle_lib as O
from common import CASE, SMALL, query_set, sha_file, sha_occ
from kmcex_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = os.path.join(ROOT, "tests", "golden", "tiny")


def test_murmur_known_answers():
    kat = {
        "ACGTACGTTGCAAGCTTAGGCTAACGTTAGC": {0: 0x40918180070a81da, 1: 0x7b518c276858d259, 6: 0x7988c8f05294cdf7, 7: 0x738c1724f7aad678,
                                            34: 0x4e630407d8fa85d2, 127: 0x32ebb6b7f970ff28},
        "T" * 31: {0: 0x378cba69158c32d6, 1: 0x36c051b9bf6fa92d, 127: 0xc41f0320aad46cb6},
        "GATTACAGATTACAGATTACAGATTACAGAT": {0: 0x26bf0b90f7af10fb, 34: 0xe4da8fc43fca375e},
        "ACGTACGTTGCAAGCTTAGGCTAACGTTAGCGGATCCATGCAATTGGCCTTAAGC": {0: 0x7fe29126d7a195ee, 7: 0x11dc98c8e8dc9ea0},
        "TGCATGCATTTTGGGGCCCCAAAATGCATGCAGGGTTTAAACCCGGGTTTACGTA": {0: 0xa05be8aa142f940a, 127: 0xc6187e097b6b887f},
    }
    for s, d in kat.items():
        for si, h in d.items():
            assert O.murmur64(s.encode(), si) == h, (s, si)
    # (k-2)-mer columns of the same table
    assert O.murmur64(b"ACGTACGTTGCAAGCTTAGGCTAACGTTAGC"[1:-1], 0) == 0x9a229d41863aebb3
    assert O.murmur64(b"ACGTACGTTGCAAGCTTAGGCTAACGTTAGC"[1:-1], 4) == 0xcc051b82e2564aaa
    assert O.murmur64(("T" * 29).encode(), 0) == 0x5536311a42c7bdc7


def test_min_kmer_known_answers():
    assert O.min_kmer("T" * 31) == "A" * 31
    assert O.min_kmer("GATTACAGATTACAGATTACAGATTACAGAT") == "ATCTGTAATCTGTAATCTGTAATCTGTAATC"
    assert O.min_kmer("TTGCATGCATTTTGGGGCCCCAAAATGCATGC") == "GCATGCATTTTGGGGCCCCAAAATGCATGCAA"
    assert O.min_kmer("TCGTNCGTTGCAAGCTTAGGCTAACGTTAGC") == "GCTAACGTTAGCCTAAGCTTGCAACGTACGA"          # N treated as A
    # k > 32 overflow (quirk Q4)
    assert O.min_kmer("ACGTACGTTGCAAGCTTAGGCTATCGTTAGCTGATCCATGCAATTGGCCTTAAGC") == "A" * 23 + "AGCTAACGA" + "T" * 23


def test_occubin_known_answers():
    b, m = O.occubin_table(1024, 7)
    for occ, (bn, mean) in {1: (1, 1), 31: (31, 31), 32: (32, 33), 34: (32, 33), 35: (33, 36), 127: (63, 126), 128: (64, 129),
                            223: (95, 222), 224: (96, 236), 248: (96, 236), 249: (97, 261), 895: (122, 886), 896: (122, 886),
                            920: (123, 911), 1000: (127, 1011), 1023: (127, 1011)}.items():
        assert (int(b[occ]), int(m[b[occ]])) == (bn, mean), occ
    b, m = O.occubin_table(4096, 9)
    for occ, (bn, mean) in {127: (127, 127), 128: (128, 129), 223: (159, 222), 224: (160, 225), 895: (383, 894), 896: (384, 908),
                            1023: (389, 1033), 4095: (511, 4083)}.items():
        assert (int(b[occ]), int(m[b[occ]])) == (bn, mean), occ
    b, m = O.occubin_table(256, 7)
    assert (int(b[224]), int(m[b[224]])) == (96, 224) and (int(b[249]), int(m[b[249]])) == (121, 249)
    with pytest.raises(ValueError):
        O.occubin_table(64, 7)                     # cs too small for nh (reference overruns its table)


def test_oracle_loads_reference_files_and_reproduces_its_answers():
    """rest.bin / km.bin written by the reference itself -> oracle query == reference kmer_to_occ."""
    m = O.OracleModel.load(TINY)
    qs = open(os.path.join(TINY, "queries.txt")).read().split()
    exp = np.loadtxt(os.path.join(TINY, "occ.txt"), dtype=np.int32)
    assert np.array_equal(m.query_strings(qs), exp)
    assert np.array_equal(m.query_packed(31, synth.from_strings(qs, 31)), exp)


@pytest.mark.parametrize("name", SMALL)
def test_oracle_build_matches_reference_hashes(name, golden, tmp_path):
    g = golden["cases"][name]
    _, k, ci, cs, nh, nb, n = CASE[name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    assert len(cnt) == g["n_kmers"]
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    d = str(tmp_path / "m")
    o.save(d)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(d, f)) == g["sha256"][f], f
    st = o.stats()
    assert (st.attempts, st.successes, st.rest_entries) == (g["stats"]["attempts"], g["stats"]["successes"], g["stats"]["rest_entries"])
    assert sha_occ(o.query_packed(k, query_set(km, k))) == g["occ_sha256"]


def test_oracle_multiblock_with_stale_slot_duplicate(golden, tmp_path):
    """2 full blocks + a partial one with unused rows: the rest table holds the Q1 duplicate rows."""
    name = "k31_multiblock_ci1"
    g = golden["cases"][name]
    _, k, ci, cs, nh, nb, n = CASE[name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    d = str(tmp_path / "m")
    o.save(d)
    assert sha_file(os.path.join(d, "km.bin")) == g["sha256"]["km.bin"]
    assert sha_file(os.path.join(d, "rest.bin")) == g["sha256"]["rest.bin"]


def test_oracle_rejects_bad_counts_and_handles_empty_filters():
    km, cnt = synth.make_stream(500, 31, 1, 1023)
    o = O.OracleModel(1, 1023, 7, 5)
    bad = cnt.copy()
    bad[3] = 2000
    with pytest.raises(RuntimeError):
        o.build(31, km, bad)
    cnt2 = np.maximum(cnt, 2).astype(np.uint32)        # empty Bloom class: the reference divides by zero at query time
    o.build(31, km, cnt2)
    assert o.stats().n_bf[0] == 0
    r = o.query_packed(31, km)
    assert len(r) == len(km)
