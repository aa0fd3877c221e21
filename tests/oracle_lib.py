"""ctypes binding of oracle/liboracle.so (the CPU checker).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")


class Stats(C.Structure):
    _fields_ = [("n_total", C.c_uint64), ("n_km", C.c_uint64), ("n_bf", C.c_uint64 * 3),
                ("attempts", C.c_uint64), ("successes", C.c_uint64), ("rest_entries", C.c_uint64),
                ("km_byte_size", C.c_uint64), ("byte_km_back", C.c_uint64),
                ("byte_bf", C.c_uint64 * 3), ("byte_bf_back", C.c_uint64 * 3)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        L = C.CDLL(LIB_PATH)
        L.kmo_murmur64.restype = C.c_uint64
        L.kmo_murmur64.argtypes = [C.c_char_p, C.c_int, C.c_uint32]
        L.kmo_hash_seed.restype = C.c_uint32
        L.kmo_min_kmer.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.kmo_occubin_table.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.kmo_create.restype = C.c_void_p
        L.kmo_create.argtypes = [C.c_int] * 4
        L.kmo_destroy.argtypes = [C.c_void_p]
        L.kmo_build.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.kmo_build_declared.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64]
        L.kmo_save.argtypes = [C.c_void_p, C.c_char_p]
        L.kmo_load.restype = C.c_void_p
        L.kmo_load.argtypes = [C.c_char_p]
        L.kmo_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.kmo_query_ascii.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_int]
        L.kmo_query_packed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        for f in ("kmo_bf", "kmo_bf_back", "kmo_value_array", "kmo_tag_array"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.kmo_km_back.restype = C.c_void_p
        L.kmo_km_back.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def murmur64(s: bytes, seed_index: int) -> int:
    L = lib()
    return L.kmo_murmur64(s, len(s), L.kmo_hash_seed(seed_index))


def min_kmer(s: str) -> str:
    out = C.create_string_buffer(len(s) + 1)
    lib().kmo_min_kmer(s.encode(), len(s), out)
    return out.raw[:len(s)].decode()


def occubin_table(max_counter: int, nh: int):
    b = np.zeros(max_counter, dtype=np.uint32)
    m = np.zeros(1 << nh, dtype=np.uint32)
    rc = lib().kmo_occubin_table(max_counter, nh, b.ctypes.data, m.ctypes.data)
    if rc:
        raise ValueError("cs too small for nh")
    return b, m


class OracleModel:
    """Mirror of the reference KModel life cycle on the CPU oracle."""

    def __init__(self, ci=1, cs=1023, nh=7, nb=5, _handle=None):
        self.L = lib()
        self.h = _handle if _handle is not None else self.L.kmo_create(ci, cs, nh, nb)
        if not self.h:
            raise ValueError("bad parameters")
        self.nb = nb

    @classmethod
    def load(cls, d: str):
        h = lib().kmo_load(d.encode())
        if not h:
            raise IOError(d)
        return cls(_handle=h)

    def build(self, k: int, kmers: np.ndarray, counts: np.ndarray, total: int | None = None):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        n = len(counts)
        rc = self.L.kmo_build(self.h, k, kmers.ctypes.data, counts.ctypes.data, n, n if total is None else total)
        if rc:
            raise RuntimeError(f"kmo_build rc={rc}")

    def build_declared(self, k: int, kmers: np.ndarray, counts: np.ndarray, n_bf, total: int):
        """The first len(counts) k-mers of a listing whose pass 1 would have found `n_bf` / `total` (kmo_build_declared)."""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        nbf = (C.c_uint64 * 3)(*[int(x) for x in (list(n_bf) + [0, 0, 0])[:3]])
        rc = self.L.kmo_build_declared(self.h, k, kmers.ctypes.data, counts.ctypes.data, len(counts), nbf, total)
        if rc:
            raise RuntimeError(f"kmo_build_declared rc={rc}")

    def array_view(self, which: str, i: int = 0) -> np.ndarray:
        """array_bytes without the copy (arrays of a 10^10-k-mer model are 3.8 GB each)"""
        st = self.stats()
        p, n = {"km_back": lambda: (self.L.kmo_km_back(self.h), st.byte_km_back), "bf": lambda: (self.L.kmo_bf(self.h, i), st.byte_bf[i]),
                "bf_back": lambda: (self.L.kmo_bf_back(self.h, i), st.byte_bf_back[i]), "value": lambda: (self.L.kmo_value_array(self.h, i), st.km_byte_size),
                "tag": lambda: (self.L.kmo_tag_array(self.h, i), st.km_byte_size)}[which]()
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(n),)) if n else np.zeros(0, np.uint8)

    def save(self, d: str):
        os.makedirs(d, exist_ok=True)
        if self.L.kmo_save(self.h, d.encode()):
            raise IOError(d)

    def stats(self) -> Stats:
        st = Stats()
        self.L.kmo_get_stats(self.h, C.byref(st))
        return st

    def query_packed(self, k: int, kmers: np.ndarray, threads: int = 8) -> np.ndarray:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        W = (k + 31) // 32
        n = kmers.size // W
        out = np.zeros(n, dtype=np.int32)
        if self.L.kmo_query_packed(self.h, k, kmers.ctypes.data, n, out.ctypes.data, threads):
            raise RuntimeError("kmo_query_packed")
        return out

    def query_strings(self, strs, threads: int = 8) -> np.ndarray:
        ln = len(strs[0])
        assert all(len(s) == ln for s in strs)
        buf = "".join(strs).encode()
        out = np.zeros(len(strs), dtype=np.int32)
        if self.L.kmo_query_ascii(self.h, buf, ln, ln, len(strs), out.ctypes.data, threads):
            raise RuntimeError("kmo_query_ascii")
        return out

    def array_bytes(self, which: str, i: int = 0) -> np.ndarray:
        st = self.stats()
        if which == "km_back":
            p, n = self.L.kmo_km_back(self.h), st.byte_km_back
        elif which == "bf":
            p, n = self.L.kmo_bf(self.h, i), st.byte_bf[i]
        elif which == "bf_back":
            p, n = self.L.kmo_bf_back(self.h, i), st.byte_bf_back[i]
        elif which == "value":
            p, n = self.L.kmo_value_array(self.h, i), st.km_byte_size
        elif which == "tag":
            p, n = self.L.kmo_tag_array(self.h, i), st.km_byte_size
        else:
            raise KeyError(which)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(n),)).copy() if n else np.zeros(0, np.uint8)

    def close(self):
        if self.h:
            self.L.kmo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def have_ref() -> bool:
    return os.path.exists(REF_DRIVER)


def ref_build(db_prefix: str, out_dir: str, ci: int, cs: int, nh: int, nb: int) -> float:
    """Runs the real reference build; returns the seconds KModel::init took (both passes + rest build)."""
    os.makedirs(out_dir, exist_ok=True)
    p = subprocess.run([REF_DRIVER, "build", db_prefix, out_dir, str(ci), str(cs), str(nh), str(nb)],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, text=True)
    for line in p.stderr.splitlines():
        if line.startswith("init_seconds"):
            return float(line.split()[1])
    return float("nan")


def ref_query(model_dir: str, strs, tmp_prefix: str, t_num: int = 8) -> np.ndarray:
    with open(tmp_prefix + ".q.txt", "w") as f:
        f.write("\n".join(strs) + "\n")
    p = subprocess.run([REF_DRIVER, "query", model_dir, tmp_prefix + ".q.txt", tmp_prefix + ".r.txt", str(t_num)],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, text=True)
    global last_ref_query_seconds
    for line in p.stderr.splitlines():
        if line.startswith("query_seconds"):
            last_ref_query_seconds = float(line.split()[1])
    return np.loadtxt(tmp_prefix + ".r.txt", dtype=np.int32, ndmin=1)


last_ref_query_seconds = float("nan")
