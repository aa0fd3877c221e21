"""CPU: the torch (device) stream generator is the same function as the numpy one."""
import numpy as np
import torch

from kmcex_amd import synth, synth_torch


def test_torch_stream_equals_numpy_stream():
    for (n, k, ci, cs, sk, sc) in [(50000, 31, 1, 1023, 1, 2), (20000, 21, 2, 255, 5, 9), (1000, 16, 1, 4095, 3, 4)]:
        km, cnt = synth.make_stream(n, k, ci, cs, seed_k=sk, seed_c=sc)
        tk, tc = synth_torch.make_stream(n, k, ci, cs, "cpu", seed_k=sk, seed_c=sc)
        assert np.array_equal(tk.numpy().view(np.uint64), km)
        assert np.array_equal(tc.numpy().view(np.uint32), cnt)


def test_revcomp_involution_and_canonical():
    for k in (31, 55, 32, 33, 21):
        km = synth.random_kmers(1000, k, seed_k=3)
        assert np.array_equal(synth.revcomp(synth.revcomp(km, k), k), km)
        s = synth.to_strings(km, k)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        rc = ["".join(comp[c] for c in reversed(x)) for x in s[:50]]
        assert synth.to_strings(synth.revcomp(km, k), k)[:50] == rc


def test_torch_stream_in_value_ranges_equals_one_shot():
    """Streams of 2^31 draws or more are generated in value ranges (torch.unique's limit); same stream."""
    dev = torch.device("cpu")
    a, ca = synth_torch.make_stream(200000, 31, 1, 1023, dev)
    b, cb = synth_torch.make_stream(200000, 31, 1, 1023, dev, range_values=30000, chunk=7777)
    assert torch.equal(a, b) and torch.equal(ca, cb)


def test_device_kmc2_writer_matches_the_numpy_writer(tmp_path):
    """bench.py's KMC2-layout writer (torch, what the init_db_kmc2 leg feeds KModel::init) writes the same two files as
    kmcex_amd.kmcdb.write_kmc2 (numpy, what the parity tests and the reference's reader are fed)"""
    import os
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from kmcex_amd import kmcdb, synth
    k, ci, cs = 31, 1, 1023
    km, cnt = synth.make_stream(60000, k, ci, cs)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    order = kmcdb.write_kmc2(a, km, cnt, k, ci, cs, n_bins=64)
    order_t = bench.write_kmc2_from_device(b, torch.from_numpy(km.view(np.int64)), torch.from_numpy(cnt.view(np.int32)), k, ci, cs, n_bins=64)
    assert np.array_equal(order, order_t.numpy())
    for ext in (".kmc_pre", ".kmc_suf"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext
