"""The CPU-side code under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool): the product's KMC
listing reader and the oracle, driven by tests/asan_driver.cpp on the committed tiny database."""
import os
import subprocess

import numpy as np

from common import sha_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kmc_reader_and_oracle_are_clean_under_asan_ubsan(tmp_path, golden):
    exe = str(tmp_path / "asan_driver")
    obj = str(tmp_path / "oracle.o")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    subprocess.check_call(["gcc", "-std=c11", "-fopenmp", "-c", os.path.join(ROOT, "oracle", "kmx_oracle.c"), "-o", obj] + san)
    subprocess.check_call(["g++", "-std=c++17", "-fopenmp", os.path.join(ROOT, "tests", "asan_driver.cpp"), os.path.join(ROOT, "kmcex_amd", "csrc", "kmc_reader.cpp"),
                           obj, "-o", exe, "-lpthread", "-lm"] + san)
    tiny = os.path.join(ROOT, "tests", "golden", "tiny")
    out = str(tmp_path / "model")
    os.makedirs(out)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    p = subprocess.run([exe, os.path.join(tiny, "db"), out, os.path.join(tiny, "queries.txt")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ERROR" not in p.stderr and "runtime error" not in p.stderr, p.stderr
    g = golden["cases"]["tiny_k31"]
    assert f"attempts {g['stats']['attempts']} successes {g['stats']['successes']} rest {g['stats']['rest_entries']}" in p.stdout
    for f in ("header", "km.bin", "rest.bin"):                              # the sanitized build writes the reference's files too
        assert sha_file(os.path.join(out, f)) == g["sha256"][f], f
    exp = np.loadtxt(os.path.join(tiny, "occ.txt"), dtype=np.int64)[:2000]
    assert f"occ_sum_first_2000 {int(exp.sum())}" in p.stdout
