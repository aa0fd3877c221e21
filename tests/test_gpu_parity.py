"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the committed goldens.

Everything here needs a real MI355X (`-m gpu`).  Bar: bit-exact (integer / byte / index work).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from common import CASE, GENOME_CASES, KMC2_CASES, LARGE, SMALL, genome_query_set, query_set, sha_file, sha_occ
from kmcex_amd import KModel, api, kmcdb, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("k", [31, 55, 21, 32, 33, 64, 16])
def test_device_hash_matches_oracle(k):
    """MurmurHash64A over the rebuilt ASCII string, k-mer and (k-2)-mer (tools.hpp:16-50)."""
    km = synth.random_kmers(2000, k, seed_k=5)
    strs = synth.to_strings(km, k)
    seeds_idx = [0, 1, 6, 7, 34, 127]
    seeds = [O.lib().kmo_hash_seed(i) for i in seeds_idx]
    for whole in (True, False):
        dev = api.debug_hash(k, km, seeds, whole)
        for i in range(0, len(strs), 37):
            s = strs[i] if whole else strs[i][1:-1]
            for j, si in enumerate(seeds_idx):
                assert int(dev[i, j]) == O.murmur64(s.encode(), si), (k, whole, strs[i], si)


def test_device_hash_known_answers():
    """Appendix C of SURVEY.md: vectors printed by the compiled reference."""
    kat = {"ACGTACGTTGCAAGCTTAGGCTAACGTTAGC": (0x40918180070a81da, 0x9a229d41863aebb3),
           "T" * 31: (0x378cba69158c32d6, 0x5536311a42c7bdc7),
           "GATTACAGATTACAGATTACAGATTACAGAT": (0x26bf0b90f7af10fb, 0xf0abc7124c776870)}
    seed0 = [O.lib().kmo_hash_seed(0)]
    for s, (h_full, h_back) in kat.items():
        km = synth.from_strings([s], 31)
        assert int(api.debug_hash(31, km, seed0, True)[0, 0]) == h_full
        assert int(api.debug_hash(31, km, seed0, False)[0, 0]) == h_back


@pytest.mark.parametrize("k", [31, 55, 21, 32, 40, 64])
def test_device_min_kmer_matches_oracle(k):
    """get_min_kmer incl. the k>32 overflow (tools.hpp:160-167, quirk Q4)."""
    km = synth.random_kmers(3000, k, seed_k=11)
    dev = api.debug_min_kmer(k, km)
    strs = synth.to_strings(km, k)
    exp = synth.from_strings([O.min_kmer(s) for s in strs], k)
    assert np.array_equal(dev.reshape(exp.shape), exp)


def test_device_modulo_is_exact_for_large_lengths():
    """`hash % length` with lengths far beyond 2^32 bits (a 10^10-k-mer model has 3.5*10^10 positions per array)."""
    rng = np.random.default_rng(3)
    h = rng.integers(0, 1 << 64, size=20000, dtype=np.uint64)
    h[:8] = [0, 1, 2**64 - 1, 2**63, 2**32, 2**32 - 1, 12345678901234567, 2**64 - 2]
    for d in [1, 2, 3, 7, 56, 2**32 - 1, 2**32, 2**32 + 1, 34999999944, 8 * 7 * 625000000, 2**40 - 87, 2**40, 2**63 - 25, 2**63]:
        dev = api.debug_mod(h, d)
        exp = np.array([int(x) % d for x in h], dtype=np.uint64)
        assert np.array_equal(dev, exp), d


def _build_both(name):
    _, k, ci, cs, nh, nb, n = CASE[name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    m = KModel(ci, cs, nh, nb)
    m.build_packed(k, km, cnt)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    return k, nb, km, cnt, m, o


def _check_arrays(m, o, nb, bf_num):
    for a in range(nb):
        assert np.array_equal(m.download("tag", a), o.array_bytes("tag", a)), f"tag array {a}"
        assert np.array_equal(m.download("value", a), o.array_bytes("value", a)), f"value array {a}"
        assert not m.download("claims", a).any(), f"claim bits left on untagged positions of array {a}"
    assert np.array_equal(m.download("km_back"), o.array_bytes("km_back"))
    for i in range(bf_num):
        assert np.array_equal(m.download("bf", i), o.array_bytes("bf", i)), f"bloom filter {i}"
        assert np.array_equal(m.download("bf_back", i), o.array_bytes("bf_back", i)), f"back filter {i}"


@pytest.mark.parametrize("name", SMALL + LARGE)
def test_build_save_query_bit_exact(name, golden, tmp_path):
    """insert -> identical arrays, stats, km.bin/rest.bin (sha256 of the REFERENCE's files) and kmer_to_occ."""
    g = golden["cases"][name]
    k, nb, km, cnt, m, o = _build_both(name)
    st, so = m.stats(), o.stats()
    assert (st.n_km, list(st.n_bf), st.km_byte_size, st.byte_km_back) == (so.n_km, list(so.n_bf), so.km_byte_size, so.byte_km_back)
    _check_arrays(m, o, nb, st.bf_num)
    assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries)
    assert st.attempts == g["stats"]["attempts"] and st.successes == g["stats"]["successes"]
    d = str(tmp_path / "model")
    os.makedirs(d)
    m.save(d)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(d, f)) == g["sha256"][f], f
    q = query_set(km, k)
    occ = m.kmer_to_occ_packed(q)
    assert np.array_equal(occ, o.query_packed(k, q))
    assert sha_occ(occ) == g["occ_sha256"]
    # reload what we saved and query again (load path)
    m2 = KModel.load(d)
    assert np.array_equal(m2.kmer_to_occ_packed(q), occ)
    # ... and a loaded model saves the same three files again (load -> save round trip)
    d2 = str(tmp_path / "model_again")
    os.makedirs(d2)
    m2.save(d2)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(d2, f)) == g["sha256"][f], f
    m.close(); m2.close(); o.close()


def test_load_reference_files_and_query_strings():
    """get_model(dir) on files written by the reference itself + the vector<string> front door."""
    d = os.path.join(ROOT, "tests", "golden", "tiny")
    m = KModel.load(d)
    qs = open(os.path.join(d, "queries.txt")).read().split()
    exp = np.loadtxt(os.path.join(d, "occ.txt"), dtype=np.int32)
    assert np.array_equal(np.array(m.kmer_to_occ(qs), dtype=np.int32), exp)
    assert m.kmer_to_occ(qs[0]) == int(exp[0])


def test_init_from_kmc_database(golden, tmp_path):
    """KModel::init(db_file): our KMC listing reader + streamed host batches -> the reference's files."""
    d = os.path.join(ROOT, "tests", "golden", "tiny")
    m = KModel(1, 1023, 7, 5)
    m.init(os.path.join(d, "db"))
    out = str(tmp_path / "m")
    os.makedirs(out)
    m.save_model(out)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(out, f)) == sha_file(os.path.join(d, f)), f


@pytest.mark.parametrize("host_decode", [0, 1], ids=["gpu_decode", "host_decode"])
def test_init_decoders_agree(host_decode, monkeypatch, tmp_path):
    """KModel::init(db): the raw-record feed decoded on the GPU (k_kmc_decode) and the host decoder (kept for databases
    with unlisted records; forced here by KMX_KMC_HOST_DECODE=1) give the resident build's arrays -- KMC1 and KMC2 layouts,
    one- and two-word k-mers, several batches of 2^22 records."""
    monkeypatch.setenv("KMX_KMC_HOST_DECODE", str(host_decode))
    for k, ci, cs, nh, nb, n, layout in ((31, 1, 1023, 7, 5, 9_000_000, "kmc1"), (55, 2, 4095, 9, 6, 300_000, "kmc2"), (23, 1, 255, 6, 3, 50_000, "kmc2")):
        km, cnt = synth.make_stream(n, k, ci, cs, seed_k=31, seed_c=32)
        db = str(tmp_path / f"db_{k}")
        if layout == "kmc1":
            kmcdb.write_kmc1(db, km, cnt, k, ci, cs)
            lk, lc = km, cnt
        else:
            order = kmcdb.write_kmc2(db, km, cnt, k, ci, cs, n_bins=5)
            lk, lc = km[order], cnt[order]
        m = KModel(ci, cs, nh, nb)
        m.init(db)
        r = KModel(ci, cs, nh, nb)
        r.build_packed(k, lk, lc)
        for a in range(nb):
            assert np.array_equal(m.download("tag", a), r.download("tag", a)) and np.array_equal(m.download("value", a), r.download("value", a))
        assert np.array_equal(m.download("km_back"), r.download("km_back")) and np.array_equal(m.download("bf", 0), r.download("bf", 0))
        sm, sr = m.stats(), r.stats()
        assert (sm.attempts, sm.successes, sm.rest_entries) == (sr.attempts, sr.successes, sr.rest_entries)
        m.close(); r.close()
    monkeypatch.delenv("KMX_KMC_HOST_DECODE")


def test_init_skips_records_outside_the_header_range(tmp_path):
    """ReadNextKmer skips records whose count lies outside the header's [min_count, max_count] (kmc_file.cpp:513); KMC
    never writes such a file, but a listing must not contain them.  Header says min_count = 3, a third of the records hold less."""
    k, cs, nh, nb = 31, 1023, 7, 5
    km, cnt = synth.make_stream(400_000, k, 1, cs, seed_k=41, seed_c=42)
    db = str(tmp_path / "db")
    kmcdb.write_kmc1(db, km, cnt, k, 3, cs)                    # every record is written, the header announces counts >= 3
    keep = cnt >= 3
    assert 0 < keep.sum() < len(cnt)
    m = KModel(3, cs, nh, nb)
    m.init(db)
    o = O.OracleModel(3, cs, nh, nb)
    o.build(k, km[keep], cnt[keep], total=len(cnt))          # KmerCount() is the header's total, listed or not (kmodel.hpp:429)
    _check_arrays(m, o, nb, 3)
    assert m.stats().rest_entries == o.stats().rest_entries


def test_streamed_ragged_batches_equal_one_shot():
    """kmx_begin / insert_batch / finish with ragged (incl. empty) batches == one-shot build."""
    name = "k31_ci2_200k"
    _, k, ci, cs, nh, nb, n = CASE[name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    m = KModel(ci, cs, nh, nb)
    n_bf = [int((cnt == ci + i).sum()) for i in range(3)]
    m.begin(k, n_bf, len(cnt))
    cuts = [0, 1, 1, 7777, 7777 + 65536, 150000, len(cnt)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        m.insert_batch(km[a:b], cnt[a:b])
    m.finish()
    _check_arrays(m, o, nb, 3)
    assert m.stats().rest_entries == o.stats().rest_entries


def test_heavy_collision_tiny_arrays():
    """Tiny coupled arrays (L = 8*(N_km>>4)*nh bits): almost every k-mer is contended, so the ordered slow
    path and the single-workgroup finisher decide the round.  Still bit-exact."""
    k, ci, cs, nh, nb = 31, 1, 1023, 7, 5
    for n in (40, 300, 3000):
        km, cnt = synth.make_stream(n, k, ci, cs, seed_k=77)
        cnt = np.maximum(cnt, 2).astype(np.uint32)          # everything into the coupled arrays
        m = KModel(ci, cs, nh, nb)
        m.build_packed(k, km, cnt)
        o = O.OracleModel(ci, cs, nh, nb)
        o.build(k, km, cnt)
        _check_arrays(m, o, nb, 0)
        st, so = m.stats(), o.stats()
        assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries)
        q = np.concatenate([km, synth.revcomp(km, k)])
        assert np.array_equal(m.kmer_to_occ_packed(q), o.query_packed(k, q))


@pytest.mark.parametrize("cfg", [(31, 2, 1023, 7, 5, 400000), (40, 1, 4095, 11, 4, 250000), (27, 1, 1023, 8, 4, 120000)],
                         ids=lambda c: "k%d_nh%d_nb%d_n%d" % (c[0], c[3], c[4], c[5]))
def test_both_finisher_paths_bit_exact(cfg, monkeypatch):
    """The single-workgroup finisher has an LDS path (sets up to 2048 / 1024 records) and a global-memory path (larger
    sets).  Both must give the oracle's arrays on the same input: KMX_FIN_GLOBAL=1 sends every set through the second;
    KMX_NSUB0/1 force the number of grid-wide passes in front of it (0: the finisher alone, from check_emit's
    snapshot; 2: resolve + reserve/resolve first) and KMX_RESOLVE_GATHER=1 the gathering form of the first pass.  The
    KMX_PIPE=0 commits the winners of every round in a launch of its own before the next check (the default lets them
    commit beside the next round's check, whose detect tables take them as settled positions).  The hooks are read when
    the model is created."""
    k, ci, cs, nh, nb, n = cfg
    km, cnt = synth.make_stream(n, k, ci, cs, seed_k=4242, seed_c=4243)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    so = o.stats()
    # KMX_NH_FIRST / KMX_NH_SECOND: the check fetches its nh positions in up to three groups and stops at the first group
    # with a conflict (default: about 2/7 and 4/7 of nh); nh / nh = all at once, 1 / 2 = one, one, the rest.
    for force_global, nsub0, gather, pipe, groups in ((0, -1, 0, 1, None), (1, -1, 0, 0, (nh, nh)), (0, 0, 0, 1, (1, 2)), (0, 2, 0, 1, None), (0, 2, 1, 0, (nh - 1, nh)),
                                                      (1, 1, 1, 1, (1, 1)), (0, -1, 0, 0, None), (0, -1, 0, 1, (nh, nh))):
        monkeypatch.setenv("KMX_PIPE", str(pipe))
        if groups:
            monkeypatch.setenv("KMX_NH_FIRST", str(groups[0]))
            monkeypatch.setenv("KMX_NH_SECOND", str(groups[1]))
        else:
            monkeypatch.delenv("KMX_NH_FIRST", raising=False)
            monkeypatch.delenv("KMX_NH_SECOND", raising=False)
        monkeypatch.setenv("KMX_FIN_GLOBAL", str(force_global))
        monkeypatch.setenv("KMX_NSUB0", str(nsub0))
        monkeypatch.setenv("KMX_NSUB1", str(nsub0))
        monkeypatch.setenv("KMX_RESOLVE_GATHER", str(gather))
        m = KModel(ci, cs, nh, nb)
        m.build_packed(k, km, cnt)
        st = m.stats()
        _check_arrays(m, o, nb, st.bf_num)
        assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries)
        assert st.contended > 0
    for v in ("KMX_FIN_GLOBAL", "KMX_NSUB0", "KMX_NSUB1", "KMX_RESOLVE_GATHER", "KMX_PIPE", "KMX_NH_FIRST", "KMX_NH_SECOND"):
        monkeypatch.delenv(v, raising=False)


@pytest.mark.parametrize("cfg", [(31, 1, 1023, 7, 5, 3000000), (40, 3, 4095, 9, 4, 600000), (27, 1, 1023, 6, 3, 2500000)], ids=lambda c: "k%d_ci%d_nh%d_n%d" % (c[0], c[1], c[3], c[5]))
def test_partitioned_bit_sets_at_every_depth(cfg, monkeypatch):
    """km_back and the Bloom slab are set by partitioned bit-sets: one tile per bin (the sizes of every other test), several
    tiles swept by one workgroup, or -- filters above 256 MB -- a second partition level (k_bs_split / k_bs_apply2).
    KMX_BS_TILE_LOG2 shrinks the tile so that a test-size filter takes each of them: 2^20 one tile, 2^13 several tiles, 2^10
    and 2^11 two levels.  The filters must be the oracle's bit for bit whichever way they were set."""
    k, ci, cs, nh, nb, n = cfg
    km, cnt = synth.make_stream(n, k, ci, cs, seed_k=777, seed_c=778)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    so = o.stats()
    for tlog2, host in ((10, 1), (11, 0), (13, 1), (20, 0), (20, 1)):
        monkeypatch.setenv("KMX_BS_TILE_LOG2", str(tlog2))
        monkeypatch.setenv("KMX_KMB_HOST", str(host))            # km_back emission riding along with the next block's late rounds, or not
        m = KModel(ci, cs, nh, nb)
        m.build_packed(k, km, cnt)
        st = m.stats()
        _check_arrays(m, o, nb, st.bf_num)
        assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries)
        del m
    monkeypatch.delenv("KMX_BS_TILE_LOG2")
    monkeypatch.delenv("KMX_KMB_HOST")


def test_finisher_alone_on_filled_arrays_is_timing_independent(monkeypatch):
    """A tiny-array configuration (nearly everything contended, sets above the finisher's LDS capacity) built 300 times
    with the finisher alone in every round (KMX_NSUB0 = KMX_NSUB1 = 0) on its global-memory path.  Up to 3 % of such
    builds used to give a different set of winners: a position that one k-mer hits with two hashes was set with one atomic
    per hash (tag|0, then tag|1) and a concurrent gatherer could see the value in between (tools/soak_case.py, DESIGN.md
    §3.1); every build must give the oracle's arrays."""
    import hashlib
    k, ci, cs, nh, nb, n, seed = 62, 3, 146, 5, 2, 120000, 834071094
    km, cnt = synth.make_stream(n, k, ci, cs, seed_k=seed, seed_c=seed + 1)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    dig = lambda x: hashlib.sha1(np.ascontiguousarray(x).tobytes()).hexdigest()
    want = [dig(o.array_bytes(w, a)) for a in range(nb) for w in ("tag", "value")] + [dig(o.array_bytes("km_back"))]
    for v, x in (("KMX_NSUB0", "0"), ("KMX_NSUB1", "0"), ("KMX_FIN_GLOBAL", "1")):
        monkeypatch.setenv(v, x)
    for it in range(300):
        m = KModel(ci, cs, nh, nb)
        m.build_packed(k, km, cnt)
        got = [dig(m.download(w, a)) for a in range(nb) for w in ("tag", "value")] + [dig(m.download("km_back"))]
        assert got == want, f"build {it} differs from the oracle"
        del m
    for v in ("KMX_NSUB0", "KMX_NSUB1", "KMX_FIN_GLOBAL"):
        monkeypatch.delenv(v)


@pytest.mark.parametrize("cfg", [(62, 3, 146, 5, 2, 120000, 834071094, 60), (31, 1, 1023, 7, 5, 3000000, 777, 25), (27, 1, 1023, 8, 4, 700000, 4242, 40)],
                         ids=lambda c: "k%d_nh%d_nb%d_n%d" % (c[0], c[3], c[4], c[5]))
def test_deferred_commit_is_timing_independent(cfg, monkeypatch):
    """The winners of a round commit INSIDE the launch that checks the next round; what that check misses reaches
    k_round_detect as settled positions.  Whether a particular tag is seen by the racing check or supplied by
    the delta differs from build to build -- the model must not: the same stream built dozens of times on the production
    path (tiny arrays where nearly every position a winner tags is claimed again a round later; three blocks; a mid-size
    case), every build compared with the oracle's arrays, and once with the commits in launches of their own (KMX_PIPE=0)."""
    import hashlib
    k, ci, cs, nh, nb, n, seed, reps = cfg
    km, cnt = synth.make_stream(n, k, ci, cs, seed_k=seed, seed_c=seed + 1)
    if n <= 700000:
        cnt = np.maximum(cnt, ci + 3).astype(np.uint32)              # everything into the coupled arrays
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    so = o.stats()
    dig = lambda x: hashlib.sha1(np.ascontiguousarray(x).tobytes()).hexdigest()
    want = [dig(o.array_bytes(w, a)) for a in range(nb) for w in ("tag", "value")] + [dig(o.array_bytes("km_back"))]
    for it in range(reps + 1):
        if it == reps:
            monkeypatch.setenv("KMX_PIPE", "0")
        m = KModel(ci, cs, nh, nb)
        m.build_packed(k, km, cnt)
        st = m.stats()
        got = [dig(m.download(w, a)) for a in range(nb) for w in ("tag", "value")] + [dig(m.download("km_back"))]
        assert got == want, f"build {it} differs from the oracle"
        assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries), f"build {it}"
        del m
    monkeypatch.delenv("KMX_PIPE")


def test_error_behaviour():
    m = KModel(1, 1023, 7, 5)
    with pytest.raises(api.KmxError):
        m.kmer_to_occ_packed(np.zeros(4, dtype=np.uint64))   # query before build
    km, cnt = synth.make_stream(1000, 31, 1, 1023)
    bad = cnt.copy()
    bad[5] = 5000                                            # count > cs: the reference indexes out of bounds
    with pytest.raises(api.KmxError):
        m.build_packed(31, km, bad)
    with pytest.raises(api.KmxError):
        m.init("/nonexistent/db")                            # reference: message + exit(1)
    with pytest.raises(api.KmxError):
        KModel.load("/nonexistent/dir")


def _random_configs(n_cfg, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n_cfg:
        k = int(rng.choice([8, 11, 15, 16, 17, 23, 31, 32, 33, 40, 47, 55, 63, 64]))
        nh = int(rng.integers(3, 17))
        nb = int(rng.choice([1, 2, 3, 5, 6, 8, 16]))
        ci = int(rng.choice([1, 2, 3]))
        e3 = 1 << nh
        cs_min = e3 // 4 + 3 * (e3 // 2) + e3 // 4          # smallest cs the OccuBin table accepts comfortably
        cs = int(min(65535, cs_min + int(rng.integers(0, 3000))))
        n = int(rng.choice([50, 700, 6000, 60000, 300000]))
        if 4 ** k < 4 * n:                                    # not enough distinct k-mers
            continue
        out.append((k, ci, cs, nh, nb, n, int(rng.integers(1, 1 << 30))))
    return out


@pytest.mark.parametrize("cfg", _random_configs(28, 20261003), ids=lambda c: "k%d_ci%d_cs%d_nh%d_nb%d_n%d" % c[:6])
def test_random_configurations_bit_exact(cfg, tmp_path):
    """Random (k, ci, cs, nh, nb, N): both template paths (nh <= 8 / > 8, one / two words), nb from 1 to 16, tiny to
    multi-list inputs, every Bloom class layout.  Arrays, statistics, files and answers must equal the oracle's."""
    k, ci, cs, nh, nb, n, seed = cfg
    km, cnt = synth.make_stream(n, k, ci, cs, seed_k=seed, seed_c=seed + 1)
    m = KModel(ci, cs, nh, nb)
    m.build_packed(k, km, cnt)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    st, so = m.stats(), o.stats()
    _check_arrays(m, o, nb, st.bf_num)
    assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries)
    d1, d2 = str(tmp_path / "g"), str(tmp_path / "o")
    os.makedirs(d1)
    m.save(d1)
    o.save(d2)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(d1, f)) == sha_file(os.path.join(d2, f)), f
    q = np.concatenate([km, synth.revcomp(km, k), synth.random_kmers(max(len(cnt) // 2, 20), k, seed_k=0xABCDEF0123)])
    assert np.array_equal(m.kmer_to_occ_packed(q), o.query_packed(k, q))
    m2 = KModel.load(d2)                                      # files written by the oracle load and answer identically
    assert np.array_equal(m2.kmer_to_occ_packed(q), o.query_packed(k, q))


@pytest.mark.parametrize("name", ["tiny_k31", "k55_nh9_nb6", "k21_nh6_nb3"])
def test_strings_the_packed_form_cannot_hold(name):
    """kmer_to_occ(vector<string>) on raw strings: 'N', lower case and other bytes are hashed as they are and count
    as 'A' only in the 2-bit conversions (tools.hpp:63-76, rest.hpp:22-34); other lengths skip the rest table
    (rest.hpp:224-226).  The byte-string kernel must agree with the oracle -- and with the packed kernel on clean input."""
    _, k, ci, cs, nh, nb, n = CASE[name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    m = KModel(ci, cs, nh, nb)
    m.build_packed(k, km, cnt)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    rng = np.random.default_rng(5)
    base = synth.to_strings(np.concatenate([km[:3000], synth.revcomp(km[3000:6000], k), synth.random_kmers(1000, k, seed_k=0xABCDEF0123)]), k)
    dirty = []
    for i, s in enumerate(base):
        if i % 3 == 0:
            dirty.append(s)                                   # clean strings inside an unclean batch
            continue
        b = list(s)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, k))] = str(rng.choice(list("NnacgtX-")))
        dirty.append("".join(b))
    got = np.array(m.kmer_to_occ(dirty), dtype=np.int32)
    assert np.array_equal(got, o.query_strings(dirty))
    clean_idx = np.arange(0, len(base), 3)
    assert np.array_equal(got[clean_idx], m.kmer_to_occ_packed(synth.from_strings([base[i] for i in clean_idx], k).reshape(-1)))
    ragged = [dirty[0], dirty[1][:-1], dirty[2][:-2], dirty[3], (dirty[4] * 2)[:min(k + 5, 64)], "AC"]
    exp = [int(o.query_strings([s])[0]) for s in ragged]
    assert m.kmer_to_occ(ragged) == exp                       # one batch, several lengths
    for L in (k - 1, k - 2, min(k + 3, 64), 2, 9):
        if L < 2 or L > 64 or L == k:
            continue
        strs = [(s * 3)[:L] for s in base[:2000]]
        assert np.array_equal(np.array(m.kmer_to_occ(strs), dtype=np.int32), o.query_strings(strs)), L


@pytest.mark.parametrize("name,k", [("tiny_k31", 31), ("k55_nh9_nb6", 55)])
def test_vector_of_strings_pipeline_over_many_chunks(name, k):
    """kmer_to_occ(vector<string>) (kmodel.hpp:90-98) on a batch of more than 3 * 2^20 strings through the three-slot pipeline
    (kmx_query_strings: tasks of 2^14 strings packed by worker threads, a chunk answered under the next one's packing): as
    separate strings and as one buffer, with strings the packed form cannot hold scattered over the batch (they are answered
    afterwards, one by one, through the byte-string kernel) -- every answer equal to the packed query's, the dirty strings'
    answers equal to the oracle's."""
    _, kk, ci, cs, nh, nb, n = CASE[name]
    assert kk == k
    km, cnt = synth.make_stream(n, k, ci, cs)
    m = KModel(ci, cs, nh, nb)
    m.build_packed(k, km, cnt)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    W = (k + 31) // 32
    nq = (1 << 20) * 3 + 12345                                  # four chunks, the last one short
    rng = np.random.default_rng(11)
    pool = np.concatenate([km.reshape(-1, W), synth.revcomp(km, k).reshape(-1, W), synth.random_kmers(len(cnt), k, seed_k=0xABCDEF0123).reshape(-1, W)])
    q = pool[rng.integers(0, len(pool), size=nq)]
    want = m.kmer_to_occ_packed(q.reshape(-1))
    rows = np.zeros((nq, 64), dtype=np.uint8)
    rows[:, :k] = synth.to_ascii(q.reshape(-1), k)
    for separate in (True, False):
        assert np.array_equal(m.kmer_to_occ_rows(rows, k, separate), want), separate
    # strings of other bytes in the second and the last quarter of the batch
    dirty_at = [(1 << 20) + 5, (1 << 20) * 2 - 1, (1 << 20) * 3 + 7, nq - 1]
    r2 = rows.copy()
    for j, i in enumerate(dirty_at):
        r2[i, (3 * j + 1) % k] = b"NnX-"[j]
    exp = want.copy()
    exp[dirty_at] = o.query_strings([bytes(r2[i, :k]).decode("latin-1") for i in dirty_at])
    for separate in (True, False):
        assert np.array_equal(m.kmer_to_occ_rows(r2, k, separate), exp), separate
    # a second model on the same thread reuses nothing of the first one's feed; a short batch after a long one reuses the slots
    assert np.array_equal(m.kmer_to_occ_rows(rows[:1000], k, True), want[:1000])
    assert np.array_equal(m.kmer_to_occ_rows(rows[:1], k, False), want[:1])


@pytest.mark.parametrize("case", KMC2_CASES, ids=lambda c: c[0])
def test_init_from_kmc2_database_unsorted_listing(case, golden, tmp_path):
    """A KMC2-layout database (what KMC 3 emits) lists bin-major, i.e. NOT in sorted order; the ordered insert must follow
    that order.  init(db) -> the files the reference wrote from the same database."""
    name, k, ci, cs, nh, nb, n, n_bins = case
    g = golden["kmc2_cases"][name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    db = str(tmp_path / "db")
    kmcdb.write_kmc2(db, km, cnt, k, ci, cs, n_bins=n_bins)
    m = KModel(ci, cs, nh, nb)
    m.init(db)
    out = str(tmp_path / "m")
    os.makedirs(out)
    m.save(out)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(out, f)) == g["sha256"][f], f
    st = m.stats()
    assert (st.attempts, st.successes, st.rest_entries) == (g["stats"]["attempts"], g["stats"]["successes"], g["stats"]["rest_entries"])
    assert sha_occ(m.kmer_to_occ_packed(query_set(km, k))) == g["occ_sha256"]


@pytest.mark.parametrize("partition", ["ring", "range"])
@pytest.mark.parametrize("name,handles", [("k31_multiblock_ci1", 2), ("k31_multiblock_ci1", 5), ("k55_multiblock", 3), ("k31_multiblock_ci2", 4), ("tiny_k31", 8),
                                          ("k31_multiblock_ci1", 1), ("k27_ci3_nb8", 3), ("k31_nh3_nb1", 3)])
def test_init_by_several_handles_from_cxx(name, handles, partition, golden, tmp_path):
    """kmx_build_from_kmc_multi_ex: KModel::init(db) by several handles from ONE process (here all on device 0), a host thread
    each.  "ring": whole arrays, peer copies.  "range" (the north star's partition): every array cut by position range, the
    words of a round written straight into the owners' inboxes, events between the steps, no host wait inside a round.
    EVERY handle ends with the reference's files; statistics and answers too."""
    from kmcex_amd import api
    _, k, ci, cs, nh, nb, n = CASE[name]
    g = golden["cases"][name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    db = str(tmp_path / "db")
    kmcdb.write_kmc1(db, km, cnt, k, ci, cs)
    ms = [KModel(ci, cs, nh, nb) for _ in range(handles)]
    api.init_multi(ms, db, partition)
    for j, m in enumerate(ms):
        out = str(tmp_path / f"m{j}")
        os.makedirs(out)
        m.save(out)
        for f in ("header", "km.bin", "rest.bin"):
            assert sha_file(os.path.join(out, f)) == g["sha256"][f], (j, f)
        st = m.stats()
        assert (st.attempts, st.successes, st.rest_entries) == (g["stats"]["attempts"], g["stats"]["successes"], g["stats"]["rest_entries"])
    assert sha_occ(ms[-1].kmer_to_occ_packed(query_set(km, k))) == g["occ_sha256"]
    api.init_multi(ms, db, partition)                               # a second build on the same handles (buffers, inboxes and events are kept or re-made)
    assert sha_occ(ms[0].kmer_to_occ_packed(query_set(km, k))) == g["occ_sha256"]
    for m in ms:
        m.close()


@pytest.mark.parametrize("name", ["k31_multiblock_ci1", "k55_multiblock", "k31_multiblock_ci2"])
def test_init_in_one_pass_gives_the_reference_files(name, golden, tmp_path, monkeypatch):
    """KMX_ONE_PASS=1 (test hook): KModel::init(db) without the host's pass 1 -- the listing is streamed to
    the device once, the classes are counted there, the build runs on the resident listing.  Same files, statistics and answers."""
    _, k, ci, cs, nh, nb, n = CASE[name]
    g = golden["cases"][name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    db = str(tmp_path / "db")
    kmcdb.write_kmc1(db, km, cnt, k, ci, cs)
    monkeypatch.setenv("KMX_TEST_HOOKS", "1")
    monkeypatch.setenv("KMX_ONE_PASS", "1")
    m = KModel(ci, cs, nh, nb)
    m.init(db)
    out = str(tmp_path / "m")
    os.makedirs(out)
    m.save(out)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(out, f)) == g["sha256"][f], f
    st = m.stats()
    assert (st.attempts, st.successes, st.rest_entries) == (g["stats"]["attempts"], g["stats"]["successes"], g["stats"]["rest_entries"])
    assert sha_occ(m.kmer_to_occ_packed(query_set(km, k))) == g["occ_sha256"]
    m.close()


@pytest.mark.parametrize("name,capx", [("k31_multiblock_ci1", None), ("k55_multiblock", None), ("k31_multiblock_ci2", None), ("k31_multiblock_ci1", "20000")])
def test_init_over_rccl_with_the_one_rank_this_box_allows(name, capx, golden, tmp_path, monkeypatch):
    """KMX_PARTITION_RANGE_RCCL: the range partition's rounds as fixed-size RCCL messages (ncclSend / ncclRecv in a group on the
    handle's stream, counts in band, librccl.so opened on demand).  RCCL wants one rank per device, so this box runs it with ONE
    handle -- every message goes to itself through RCCL; the reference's files must come out.  capx: regions forced far too small
    (test hook) -- words are dropped, the build is void and is repeated through the inboxes: the files are still the reference's.
    Two handles on one device are refused with a message."""
    from kmcex_amd import api
    _, k, ci, cs, nh, nb, n = CASE[name]
    g = golden["cases"][name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    db = str(tmp_path / "db")
    kmcdb.write_kmc1(db, km, cnt, k, ci, cs)
    if capx:
        monkeypatch.setenv("KMX_TEST_HOOKS", "1")
        monkeypatch.setenv("KMX_RANGE_CAPX", capx)
    m = KModel(ci, cs, nh, nb)
    api.init_multi([m], db, "range-rccl")
    out = str(tmp_path / "m")
    os.makedirs(out)
    m.save(out)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(out, f)) == g["sha256"][f], f
    st = m.stats()
    assert (st.attempts, st.successes, st.rest_entries) == (g["stats"]["attempts"], g["stats"]["successes"], g["stats"]["rest_entries"])
    assert sha_occ(m.kmer_to_occ_packed(query_set(km, k))) == g["occ_sha256"]
    m2 = KModel(ci, cs, nh, nb)
    with pytest.raises(api.KmxError, match="one handle per device"):
        api.init_multi([m, m2], db, "range-rccl")
    m.close()
    m2.close()


@pytest.mark.parametrize("partition", ["ring", "range"])
def test_init_by_several_handles_unsorted_listing(partition, golden, tmp_path):
    """the same on a KMC2-layout database (bin-major listing): the slices of the ranks follow the listing, not the sorted order"""
    from kmcex_amd import api
    name, k, ci, cs, nh, nb, n, n_bins = KMC2_CASES[0]
    g = golden["kmc2_cases"][name]
    km, cnt = synth.make_stream(n, k, ci, cs)
    db = str(tmp_path / "db")
    kmcdb.write_kmc2(db, km, cnt, k, ci, cs, n_bins=n_bins)
    ms = [KModel(ci, cs, nh, nb) for _ in range(3)]
    api.init_multi(ms, db, partition)
    out = str(tmp_path / "m")
    os.makedirs(out)
    ms[1].save(out)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(out, f)) == g["sha256"][f], f


@pytest.mark.parametrize("case", GENOME_CASES, ids=lambda c: c[0])
def test_genome_like_stream_neighbour_path(case, golden, tmp_path):
    """Overlapping k-mers of a sequence: the de Bruijn neighbours of a stored k-mer are stored too, so the query's
    neighbour-based disambiguation (kmodel.hpp:286-359) is exercised for real.  Files and answers = the reference's."""
    name, k, ci, cs, nh, nb, n_bases = case
    g = golden["genome_cases"][name]
    km, cnt = synth.genome_stream(n_bases, k, ci, cs)
    assert len(cnt) == g["n_kmers"]
    m = KModel(ci, cs, nh, nb)
    m.build_packed(k, km, cnt)
    out = str(tmp_path / "m")
    os.makedirs(out)
    m.save(out)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(out, f)) == g["sha256"][f], f
    q = genome_query_set(km, k)
    occ = m.kmer_to_occ_packed(q)
    assert sha_occ(occ) == g["occ_sha256"]
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    assert np.array_equal(occ, o.query_packed(k, q))
    strs = synth.to_strings(q[:20000], k)                       # the raw-string kernel takes the same neighbour path
    dirty = [s if i % 2 else s[:7] + "N" + s[8:] for i, s in enumerate(strs)]
    assert np.array_equal(np.array(m.kmer_to_occ(dirty), dtype=np.int32), o.query_strings(dirty))


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 33, 100])
def test_degenerate_sizes(n, tmp_path):
    """Empty and tiny listings: zero-length filters (N_km < 16, N_bf < 8) follow the documented divergence D2 --
    nothing is stored in an empty filter, every probe of it misses -- identically in the oracle and on the GPU."""
    k, ci, cs, nh, nb = 31, 1, 1023, 7, 5
    km, cnt = synth.make_stream(max(n, 1), k, ci, cs, seed_k=99)
    km, cnt = km[:n], cnt[:n]
    m = KModel(ci, cs, nh, nb)
    m.build_packed(k, km, cnt)
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    st, so = m.stats(), o.stats()
    assert (st.n_total, st.n_km, st.km_byte_size, st.rest_entries) == (so.n_total, so.n_km, so.km_byte_size, so.rest_entries)
    _check_arrays(m, o, nb, st.bf_num)
    d1, d2 = str(tmp_path / "g"), str(tmp_path / "o")
    os.makedirs(d1)
    m.save(d1)
    o.save(d2)
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(d1, f)) == sha_file(os.path.join(d2, f)), f
    q = np.concatenate([km, synth.random_kmers(50, k, seed_k=0xABCDEF0123)])
    assert np.array_equal(m.kmer_to_occ_packed(q), o.query_packed(k, q))
    assert np.array_equal(KModel.load(d1).kmer_to_occ_packed(q), o.query_packed(k, q))


def test_api_state_machine():
    """Calls out of order or with unusable arguments return an error code; nothing crashes, nothing falls back."""
    k, ci, cs, nh, nb = 31, 1, 1023, 7, 5
    km, cnt = synth.make_stream(5000, k, ci, cs)
    m = KModel(ci, cs, nh, nb)
    with pytest.raises(api.KmxError):
        m.finish()                                            # finish before begin
    with pytest.raises(api.KmxError):
        m.insert_batch(km, cnt)                               # insert before begin
    with pytest.raises(api.KmxError):
        m.save("/tmp")                                        # save before build
    m.begin(k, [int((cnt == 1).sum())], len(cnt))
    m.insert_batch(km[:0], cnt[:0])                           # empty batch is fine
    with pytest.raises(api.KmxError):
        m.kmer_to_occ_packed(km)                              # query while building
    m.insert_batch(km, cnt)
    m.finish()
    with pytest.raises(api.KmxError):
        m.save("/nonexistent/dir/x")                          # the reference needs an existing directory too (main.cpp:148)
    with pytest.raises(api.KmxError):
        m.kmer_to_occ(["A"])                                  # shorter than 2 characters
    with pytest.raises(api.KmxError):
        api.KModel(1, 1023, 2, 5)                             # nh out of range
    with pytest.raises(api.KmxError):
        api.KModel(1, 100, 7, 5)                              # cs too small for nh=7 (occu_bin.hpp:38-44 overruns)
    m.build_packed(k, km, cnt)                                # a handle can be rebuilt
    o = O.OracleModel(ci, cs, nh, nb)
    o.build(k, km, cnt)
    assert np.array_equal(m.kmer_to_occ_packed(km), o.query_packed(k, km))
    m.close()
    m.close()                                                 # idempotent
