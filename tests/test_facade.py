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


@pytest.mark.gpu
@pytest.mark.parametrize("devices,partition", [("0,0", "ring"), ("0,0,0", "ring"), ("0,0,0,0,0", "ring"), ("0", "range"), ("0,0", "range"), ("0,0,0", "range"), ("0,0,0,0,0,0,0,0", "range"), ("0", "range-rccl")])
def test_driver_on_several_devices_reproduces_reference_model_files(tmp_path, devices, partition):
    """KMX_DEVICES + KMX_PARTITION: KModel::init from C++ on several handles (all on device 0 here: the pool has one GPU) -- one
    host thread per handle; the ring of whole arrays with hipMemcpyPeerAsync hand-offs, or the north star's position-range
    partition with the words of a round written into the owners' inboxes (kmx_build_from_kmc_multi_ex) -- writes the reference's files."""
    exe = _compile(tmp_path)
    tiny = os.path.join(ROOT, "tests", "golden", "tiny")
    env = dict(os.environ, KMC_BIN="/nonexistent", KMX_DEVICES=devices, KMX_PARTITION=partition)
    p = subprocess.run([exe, "-k31", "-nh7", "-nb5", "-ci1", "-cs1023", "reads.fq", os.path.join(tiny, "db"), str(tmp_path)],
                       capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    for f in ("header", "km.bin", "rest.bin"):
        assert sha_file(os.path.join(str(tmp_path), "db", f)) == sha_file(os.path.join(tiny, f)), f


@pytest.mark.gpu
@pytest.mark.parametrize("var,value", [("KMX_DEVICES", "0,,1"), ("KMX_DEVICES", "0,x"), ("KMX_DEVICES", "0,"), ("KMX_PARTITION", "hash")])
def test_driver_rejects_settings_that_do_not_parse(tmp_path, var, value):
    """a malformed KMX_DEVICES / KMX_PARTITION is an error message and exit(1), not a silent single-GPU build"""
    exe = _compile(tmp_path)
    tiny = os.path.join(ROOT, "tests", "golden", "tiny")
    env = dict(os.environ, KMC_BIN="/nonexistent", KMX_DEVICES="0,0")
    env[var] = value
    p = subprocess.run([exe, "-k31", "-nh7", "-nb5", "-ci1", "-cs1023", "reads.fq", os.path.join(tiny, "db"), str(tmp_path)],
                       capture_output=True, text=True, env=env)
    assert p.returncode == 1 and var in p.stdout, p.stdout + p.stderr


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


REF_MAIN = "/root/reference/main.cpp"
FACADE_EXE = os.path.join(ROOT, "oracle", "_ref", "kmcEx_facade")


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference is only present in the build container")
def test_reference_main_cpp_compiles_against_the_facade(tmp_path):
    """The reference's own caller (main.cpp:11-13, 56-62, 74, 155) compiles UNMODIFIED against include/kmodel.hpp with the
    reference's flags (makefile:4) and links libkmx.so.  The copy lives in tmp_path only; next to it there is no
    reference kmodel.hpp, so `#include "kmodel.hpp"` can only resolve to the facade."""
    import shutil
    api.load_library()
    src = str(tmp_path / "main.cpp")
    shutil.copyfile(REF_MAIN, src)
    exe = str(tmp_path / "kmcEx")
    subprocess.check_call(["g++", "-O3", "-m64", "-fopenmp", "-std=c++11", "-w", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + os.path.join(ROOT, "kmcex_amd"), "-lkmx", "-Wl,-rpath," + os.path.join(ROOT, "kmcex_amd"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert "kmcEx [options] <input_file_name> <output_file_name> <working_directory>" in p.stdout      # read_me(), main.cpp:30-55
    # the committed recipe builds the same thing into oracle/_ref (it travels to the GPU box; the source does not)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "facade"])
    assert os.path.exists(FACADE_EXE)


@pytest.mark.gpu
def test_reference_main_cpp_binary_reproduces_reference_model_files(tmp_path):
    """oracle/_ref/kmcEx_facade = /root/reference/main.cpp compiled against the facade (oracle/Makefile `facade`): run the
    way the README runs kmcEx, on the tiny golden database, it writes the reference's three files.  (main.cpp:140 shells
    out to ./kmc_api/kmc, absent like in the reference snapshot; `run` ignores that and opens the database that is there.
    main.cpp:150 has no return statement, so the exit status is not the reference's to define: only the files are compared.)"""
    if not os.path.exists(FACADE_EXE):
        pytest.skip("oracle/_ref/kmcEx_facade was not built (no reference at build time)")
    import shutil
    tiny = os.path.join(ROOT, "tests", "golden", "tiny")
    work = tmp_path / "work"
    work.mkdir()
    for ext in (".kmc_pre", ".kmc_suf"):
        shutil.copyfile(os.path.join(tiny, "db" + ext), str(tmp_path / ("db" + ext)))
    p = subprocess.run([FACADE_EXE, "-k31", "-nh7", "-nb5", "-ci1", "-cs1023", "reads.fq", str(tmp_path / "db"), str(work)],
                       capture_output=True, text=True, cwd=str(tmp_path))
    out = work / "db"
    for f in ("header", "km.bin", "rest.bin"):
        assert os.path.exists(str(out / f)), p.stdout[-500:] + p.stderr[-500:]
        assert sha_file(str(out / f)) == sha_file(os.path.join(tiny, f)), f
