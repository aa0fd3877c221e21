"""CPU: the C-ABI library loads, exports every symbol include/kmx.h declares, and fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from kmcex_amd import api, kmcdb, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "kmx.h")).read()
    return sorted(set(re.findall(r"\b(kmx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_agree():
    assert _declared() == sorted(api.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = api.load_library()
    raw = ctypes.CDLL(api.lib_path())
    for s in _declared():
        assert hasattr(raw, s), s
    assert L.kmx_last_error() is not None


def test_struct_layouts_match_the_header(tmp_path):
    """kmx_stats and kmx_ring_list as gcc lays them out from include/kmx.h have the size of their ctypes mirrors (a mirror
    that is too small lets kmx_get_stats write past it)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "kmx.h"\nint main(void){ printf("%zu %zu %d\\n", sizeof(kmx_stats), sizeof(kmx_ring_list), KMX_KERNEL_CLASSES); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    a, b, c = (int(x) for x in subprocess.check_output([str(exe)]).split())
    assert a == ctypes.sizeof(api.Stats) and b == ctypes.sizeof(api.RingList)
    assert c == len(api.KModel.KERNEL_CLASSES)


def test_occubin_host_table_matches_oracle():
    import oracle_lib as O
    for cs, nh in ((1023, 7), (4095, 9), (255, 7), (1023, 6)):
        b, m = api.occubin(cs, nh)
        ob, om = O.occubin_table(cs + 1, nh)
        assert np.array_equal(b, ob) and np.array_equal(m, om)
    with pytest.raises(api.KmxError):
        api.occubin(63, 7)


def test_kmc_reader_lists_what_the_writer_wrote(tmp_path):
    """Our KMC listing reader (host C++, in libkmx.so) against the numpy KMC1 writer; no GPU involved."""
    for k, ci, cs, n in ((31, 1, 1023, 30000), (55, 2, 4095, 5000), (21, 1, 255, 2000), (32, 1, 65535, 1000)):
        km, cnt = synth.make_stream(n, k, ci, cs)
        db = str(tmp_path / f"db{k}")
        kmcdb.write_kmc1(db, km, cnt, k, ci, cs)
        k2, total, okm, ocnt = api.kmc_list(db)
        assert (k2, total) == (k, len(cnt))
        assert np.array_equal(okm.reshape(km.shape), km) and np.array_equal(ocnt, cnt)
    # counts outside [min_count, max_count] of the header are skipped, total_kmers is not (kmc_file.cpp:513, :763)
    km, cnt = synth.make_stream(1000, 31, 1, 1023)
    cnt2 = cnt.copy()
    cnt2[::10] = 2000
    db = str(tmp_path / "filtered")
    kmcdb.write_kmc1(db, km, cnt2, 31, 1, 1023, counter_size=2)
    _, total, okm, ocnt = api.kmc_list(db)
    keep = cnt2 <= 1023
    assert total == 1000 and np.array_equal(okm, km[keep]) and np.array_equal(ocnt, cnt2[keep])
    with pytest.raises(api.KmxError):
        api.kmc_list(str(tmp_path / "missing"))


def test_kmc_reader_record_shapes_and_slice_boundaries(tmp_path):
    """Every record shape the decoder has a fast path for (suffix of 1..8 bytes, counters of 1..3 bytes, k from 9 to 35
    incl. the two-word k = 33..35 whose LUT prefix straddles the words) and sizes around the per-thread slice boundaries
    (1, 2, a few, 65536 +- 1, several slices): the listing equals what the writer wrote, also when the last records of a
    slice take the byte-wise tail path."""
    rng = np.random.default_rng(5)
    for k in (9, 12, 13, 16, 21, 27, 31, 32, 33, 35):
        for cs in (255, 1023, 70000):
            for n in (1, 2, 7, 65535, 65537, 200001):
                if 4 ** k < 4 * n:
                    continue
                if rng.random() < 0.5 and n > 1000:           # half of the big ones: enough to cover every shape
                    continue
                km, cnt = synth.make_stream(n, k, 1, cs, seed_k=int(rng.integers(1, 1 << 20)))
                db = str(tmp_path / f"db_{k}_{cs}_{n}")
                kmcdb.write_kmc1(db, km, cnt, k, 1, cs)
                k2, total, okm, ocnt = api.kmc_list(db)
                assert (k2, total) == (k, len(cnt)), (k, cs, n)
                assert np.array_equal(okm.reshape(km.shape), km) and np.array_equal(ocnt, cnt), (k, cs, n)
                os.remove(db + ".kmc_pre"); os.remove(db + ".kmc_suf")


def test_kmc2_layout_lists_bin_major(tmp_path):
    """KMC2 prefix files (per-bin LUTs, kmc_file.cpp:188-235, :449): the listing walks bin after bin."""
    for k, ci, cs, n, n_bins in ((31, 1, 1023, 20000, 5), (55, 1, 4095, 4000, 3), (21, 2, 255, 2500, 16)):
        km, cnt = synth.make_stream(n, k, ci, cs)
        db = str(tmp_path / f"k2_{k}")
        order = kmcdb.write_kmc2(db, km, cnt, k, ci, cs, n_bins=n_bins)
        k2, total, okm, ocnt = api.kmc_list(db)
        exp = km.reshape(len(cnt), -1)[order]
        assert (k2, total) == (k, len(cnt))
        assert np.array_equal(okm.reshape(exp.shape), exp) and np.array_equal(ocnt, cnt[order])
        assert not np.array_equal(order, np.arange(len(order)))          # really not the sorted order


def test_tiny_golden_database_lists_in_order():
    k, total, km, cnt = api.kmc_list(os.path.join(ROOT, "tests", "golden", "tiny", "db"))
    assert k == 31 and total == len(cnt) == 20000
    assert np.all(km[1:] > km[:-1])


@pytest.mark.skipif(api.device_count() > 0, reason="this check is for machines without a GPU")
def test_no_gpu_means_loud_failure_not_a_fallback():
    with pytest.raises(api.KmxError) as e:
        api.KModel(1, 1023, 7, 5)
    assert e.value.code == -2
    with pytest.raises(api.KmxError):
        api.KModel.load(os.path.join(ROOT, "tests", "golden", "tiny"))


def test_string_packer_matches_numpy_at_every_length():
    """strpack.cpp (the host half of kmer_to_occ(vector<string>), kmodel.hpp:90-98): 16 characters per SSSE3 step, overlapping
    loads, two-word k-mers -- against synth.to_ascii's inverse, as one buffer and as separate strings; no GPU involved."""
    rng = np.random.default_rng(5)
    for ln in list(range(2, 65)):
        n = 300
        km = synth.random_kmers(n, ln, seed_k=1000 + ln)
        rows = np.zeros((n, 72), dtype=np.uint8)
        rows[:, :ln] = synth.to_ascii(km, ln)
        rows[:, ln:] = rng.integers(0, 256, size=(n, 72 - ln), dtype=np.uint8)       # what follows a string is not looked at
        for separate in (False, True):
            got, clean = api.pack_strings_len(rows, ln, separate)
            assert clean and np.array_equal(got, np.ascontiguousarray(km).reshape(-1)), (ln, separate)
        for bad in (b"N", b"a", b"\0", b"U"):
            r2 = rows.copy()
            r2[int(rng.integers(n)), int(rng.integers(ln))] = bad[0]
            assert not api.pack_strings_len(r2, ln, bool(ln & 1))[1], (ln, bad)


def test_kernel_class_count_is_exported():
    L = api.load_library()
    assert L.kmx_kernel_classes() == len(api.KModel.KERNEL_CLASSES)
