"""The C++ facade (include/kmodel.hpp) and the main.cpp-equivalent driver: they compile with the reference's own
flags (g++ -std=c++11, makefile:4) and link libkmx.so; on a GPU the driver reproduces the reference's files."""
import os
import subprocess

import pytest

from common import sha_file
from kmcex_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path, source=os.path.join(ROOT, "examples", "kmcex_main.cpp"), name="kmcEx"):
    api.load_library()
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-O3", "-m64", "-std=c++11", "-I" + os.path.join(ROOT, "include"), source,
                           "-L" + os.path.join(ROOT, "kmcex_amd"), "-lkmx", "-Wl,-rpath," + os.path.join(ROOT, "kmcex_amd"), "-o", exe])
    return exe


def test_facade_compiles_as_cxx11_and_links(tmp_path):
    exe = _compile(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "USAGE" in p.stdout          # too few arguments -> usage, like read_me() (main.cpp:30-55)


@pytest.mark.gpu
def test_driver_reproduces_reference_model_files(tmp_path):
    exe = _compile(tmp_path)
    tiny = os.path.join(ROOT, "tests", "golden", "tiny")
    env = dict(os.environ, KMC_BIN="/nonexistent")
    p = subprocess.run([exe, "-k31", "-nh7", "-nb5", "-ci1", "-cs1023", "reads.fq", os.path.join(tiny, "db"), str(tmp_path)],
                       capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(str(tmp_path), "db", f)) == sha_file(os.path.join(tiny, f)), f


def test_facade_query_program_compiles(tmp_path):
    _compile(tmp_path, os.path.join(ROOT, "tests", "facade_query.cpp"), "facade_query")


@pytest.mark.gpu
def test_facade_vector_of_strings_gives_reference_answers(tmp_path):
    """get_model(dir) + kmer_to_occ(vector<string>) through include/kmodel.hpp on the REFERENCE's model files and query
    strings (tests/golden/tiny): the printed answers are the reference's occ.txt, line for line."""
    exe = _compile(tmp_path, os.path.join(ROOT, "tests", "facade_query.cpp"), "facade_query")
    tiny = os.path.join(ROOT, "tests", "golden", "tiny")
    p = subprocess.run([exe, tiny, os.path.join(tiny, "queries.txt")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-300:] + p.stderr
    assert p.stdout.split() == open(os.path.join(tiny, "occ.txt")).read().split()
