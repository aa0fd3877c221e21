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
