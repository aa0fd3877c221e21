"""Python mirror of the reference's KModel interface (kmodel.hpp) over the C ABI of libkmx.so.

Names follow the reference / its README: ``get_model``, ``init`` (``init_KModel``), ``kmer_to_occ``,
``save`` (``save_model``), ``load`` (``load_model``).  All compute happens in the HIP library; this module
only marshals pointers.  It raises ``KmxError`` (there is no CPU fallback) when the library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

_PKG = os.path.dirname(os.path.abspath(__file__))


class KmxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"kmx error {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [("n_total", C.c_uint64), ("n_km", C.c_uint64), ("n_bf", C.c_uint64 * 3),
                ("attempts", C.c_uint64), ("successes", C.c_uint64), ("rest_entries", C.c_uint64),
                ("km_byte_size", C.c_uint64), ("byte_km_back", C.c_uint64),
                ("byte_bf", C.c_uint64 * 3), ("byte_bf_back", C.c_uint64 * 3),
                ("fast_commits", C.c_uint64), ("contended", C.c_uint64), ("finisher_iters", C.c_uint64),
                ("blocks", C.c_uint64), ("rounds", C.c_uint64),
                ("k", C.c_int32), ("ci", C.c_int32), ("cs", C.c_int32), ("nh", C.c_int32), ("nb", C.c_int32),
                ("bf_num", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32), ("rest_bytes", C.c_uint64),
                ("piped_attempts", C.c_uint64), ("piped_commits", C.c_uint64),
                ("piped_gathers", C.c_uint64), ("piped_atomics", C.c_uint64),
                ("query_neighbour_calls", C.c_uint64), ("query_accounted", C.c_uint64)]


# every symbol include/kmx.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "kmx_last_error", "kmx_device_count", "kmx_create", "kmx_destroy", "kmx_set_stream", "kmx_build_from_kmc",
    "kmx_begin", "kmx_insert_batch", "kmx_insert_batch_dev", "kmx_finish", "kmx_build_dev", "kmx_build_host",
    "kmx_query_packed", "kmx_query_packed_dev", "kmx_query_ascii", "kmx_query_strings", "kmx_save", "kmx_load", "kmx_get_stats",
    "kmx_download", "kmx_debug_hash", "kmx_debug_min_kmer", "kmx_occubin", "kmx_microbench", "kmx_last_build_seconds",
    "kmx_set_profile", "kmx_get_kernel_times", "kmx_kmc_info", "kmx_kmc_read", "kmx_debug_mod",
    "kmx_count_classes_dev", "kmx_shard_begin", "kmx_shard_classify_dev", "kmx_ring_msg_bytes", "kmx_ring_round_dev",
    "kmx_ring_stale_dup_dev", "kmx_shard_local", "kmx_shard_complete", "kmx_dev_view", "kmx_or_words_dev",
    "kmx_debug_pack_strings", "kmx_kernel_classes", "kmx_abi_version", "kmx_get_stats_n",
    "kmx_create_on", "kmx_build_from_kmc_multi", "kmx_build_from_kmc_multi_ex", "kmx_range_begin", "kmx_range_buffers", "kmx_range_emit_dev", "kmx_range_verdict_dev", "kmx_range_resolve_dev", "kmx_range_commit_dev", "kmx_range_flush_dev", "kmx_range_inband", "kmx_range_verdict_inband_dev", "kmx_range_commit_inband_dev",
]


class RingList(C.Structure):
    """kmx_ring_list of include/kmx.h"""
    _fields_ = [("list", C.c_int32), ("n_host", C.c_int32), ("src_kmers", C.c_void_p), ("src_counts", C.c_void_p),
                ("src_msg", C.c_void_p), ("dst_msg", C.c_void_p)]

_lib = None


def lib_path() -> str:
    return _build.LIB


def load_library():
    """dlopen libkmx.so (building it first if hipcc is available and the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    # a specially built variant (tools/stress_small_tables.py): a test hook like the library's own, honoured only with KMX_TEST_HOOKS=1
    path = os.environ.get("KMX_LIBRARY") if os.environ.get("KMX_TEST_HOOKS") == "1" else None
    if os.environ.get("KMX_LIBRARY") and path is None:
        # results must not be taken for the variant's: refuse instead of silently running the stock library
        raise KmxError(-2, "KMX_LIBRARY is set but KMX_TEST_HOOKS=1 is not: the variant library would be ignored")
    if path:
        if not os.path.exists(path):
            raise KmxError(-2, f"KMX_LIBRARY={path} does not exist")
    else:
        path = _build.LIB
    if path == _build.LIB and (not os.path.exists(path) or _build.stale()):
        try:
            _build.build_lib()
        except Exception as e:  # noqa: BLE001
            # never run against a library older than the sources: the numbers would describe code that is not in the tree
            what = "is missing" if not os.path.exists(path) else "is older than its sources"
            raise KmxError(-2, f"libkmx.so {what} and could not be built ({e}); there is no CPU fallback")
    # A process that also uses PyTorch-ROCm must end up with ONE HIP runtime: torch bundles its own libamdhip64
    # (same SONAME as /opt/rocm's).  Importing torch first makes libkmx.so bind to the copy torch already loaded;
    # the other order leaves torch unable to see the GPU ("No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.kmx_last_error.restype = C.c_char_p
    L.kmx_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    L.kmx_destroy.argtypes = [vp]
    L.kmx_set_stream.argtypes = [vp, vp]
    L.kmx_build_from_kmc.argtypes = [vp, C.c_char_p]
    L.kmx_begin.argtypes = [vp, i32, C.POINTER(u64), u64]
    L.kmx_insert_batch.argtypes = [vp, vp, vp, u64]
    L.kmx_insert_batch_dev.argtypes = [vp, vp, vp, u64]
    L.kmx_finish.argtypes = [vp]
    L.kmx_build_dev.argtypes = [vp, i32, vp, vp, u64]
    L.kmx_build_host.argtypes = [vp, i32, vp, vp, u64]
    L.kmx_query_packed.argtypes = [vp, vp, u64, vp]
    L.kmx_query_packed_dev.argtypes = [vp, vp, u64, vp]
    L.kmx_query_ascii.argtypes = [vp, C.c_char_p, i32, i32, u64, vp]
    L.kmx_query_strings.argtypes = [vp, C.POINTER(C.c_char_p), i32, u64, vp]
    L.kmx_save.argtypes = [vp, C.c_char_p]
    L.kmx_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.kmx_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.kmx_download.argtypes = [vp, i32, i32, vp, u64, C.POINTER(u64)]
    L.kmx_debug_hash.argtypes = [i32, vp, u64, vp, i32, i32, vp]
    L.kmx_debug_min_kmer.argtypes = [i32, vp, u64, vp]
    L.kmx_occubin.argtypes = [i32, i32, vp, vp]
    L.kmx_debug_mod.argtypes = [vp, u64, u64, vp]
    L.kmx_microbench.argtypes = [i32, u64, u64, i32, C.POINTER(C.c_double)]
    L.kmx_last_build_seconds.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.kmx_kmc_info.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(u64)]
    L.kmx_kmc_read.argtypes = [C.c_char_p, vp, vp, u64, C.POINTER(u64)]
    L.kmx_count_classes_dev.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.kmx_shard_begin.argtypes = [vp, i32, C.POINTER(u64), u64, i32, i32]
    L.kmx_shard_classify_dev.argtypes = [vp, vp, vp, u64, vp, vp, C.POINTER(u64)]
    L.kmx_ring_msg_bytes.restype = u64
    L.kmx_ring_msg_bytes.argtypes = [i32]
    L.kmx_ring_round_dev.argtypes = [vp, i32, C.POINTER(RingList), i32]
    L.kmx_ring_stale_dup_dev.argtypes = [vp, i32]
    L.kmx_shard_local.argtypes = [vp, C.POINTER(Stats), C.POINTER(vp), C.POINTER(vp)]
    L.kmx_shard_complete.argtypes = [vp, vp, vp, u64, C.POINTER(Stats)]
    L.kmx_dev_view.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(u64)]
    _sig(L, "kmx_create_on", [i32, i32, i32, i32, i32, C.POINTER(vp)])
    _sig(L, "kmx_build_from_kmc_multi", [C.POINTER(vp), i32, C.c_char_p])
    _sig(L, "kmx_build_from_kmc_multi_ex", [C.POINTER(vp), i32, C.c_char_p, i32])
    _sig(L, "kmx_range_begin", [vp, i32, C.POINTER(u64), u64, i32, i32])
    _sig(L, "kmx_range_buffers", [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)])
    _sig(L, "kmx_range_emit_dev", [vp, i32, C.POINTER(RingList), i32, C.POINTER(u64)])
    _sig(L, "kmx_range_verdict_dev", [vp, i32, vp, C.POINTER(u64), C.POINTER(u64), i32, vp])
    _sig(L, "kmx_range_resolve_dev", [vp, i32, vp])
    _sig(L, "kmx_range_commit_dev", [vp, vp, u64])
    _sig(L, "kmx_range_flush_dev", [vp, C.POINTER(u64)])
    _sig(L, "kmx_range_inband", [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)])
    _sig(L, "kmx_range_verdict_inband_dev", [vp, i32, vp, i32, vp])
    _sig(L, "kmx_range_commit_inband_dev", [vp, vp, i32])
    L.kmx_or_words_dev.argtypes = [vp, vp, vp, u64]
    _sig(L, "kmx_debug_pack_strings", [vp, vp, i32, i32, u64, vp, C.POINTER(i32)])
    _sig(L, "kmx_kernel_classes", [])
    _sig(L, "kmx_abi_version", [])
    _sig(L, "kmx_get_stats_n", [vp, vp, u64])
    L.kmx_set_profile.argtypes = [vp, i32]
    L.kmx_get_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64), i32]
    _lib = L
    return L


def _sig(L, name, argtypes):
    """argtypes of an entry point a variant library (KMX_LIBRARY: an earlier round's build, for same-box comparisons) may lack"""
    f = getattr(L, name, None)
    if f is not None:
        f.argtypes = argtypes


def _chk(rc: int):
    if rc != 0:
        raise KmxError(rc, load_library().kmx_last_error().decode(errors="replace"))


def device_count() -> int:
    return load_library().kmx_device_count()


def occubin(cs: int, nh: int):
    b = np.zeros(cs + 1, dtype=np.uint32)
    m = np.zeros(1 << nh, dtype=np.uint32)
    _chk(load_library().kmx_occubin(cs, nh, b.ctypes.data, m.ctypes.data))
    return b, m


def debug_hash(k: int, kmers: np.ndarray, seeds, whole: bool = True) -> np.ndarray:
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    n = kmers.size // ((k + 31) // 32)
    out = np.zeros((n, len(seeds)), dtype=np.uint64)
    _chk(load_library().kmx_debug_hash(k, kmers.ctypes.data, n, seeds.ctypes.data, len(seeds), int(whole), out.ctypes.data))
    return out


def debug_min_kmer(k: int, kmers: np.ndarray) -> np.ndarray:
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
    n = kmers.size // ((k + 31) // 32)
    out = np.zeros_like(kmers)
    _chk(load_library().kmx_debug_min_kmer(k, kmers.ctypes.data, n, out.ctypes.data))
    return out


def debug_mod(h: np.ndarray, d: int) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.uint64)
    out = np.zeros_like(h)
    _chk(load_library().kmx_debug_mod(h.ctypes.data, len(h), d, out.ctypes.data))
    return out


def pack_strings(flat: np.ndarray, separate: bool = False):
    """Host half of the vector<string> front door on its own (strpack.cpp): uint8[n, stride >= len] rows -> (packed u64[n * W], clean).
    `separate`: hand the rows over as n pointers (what a vector<string> holds) instead of one buffer."""
    L = load_library()
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    n, stride = flat.shape
    return _pack(L, flat, n, stride, stride, separate)


def _pack(L, flat, n, ln, stride, separate):
    out = np.zeros(n * ((ln + 31) // 32), dtype=np.uint64)
    clean = C.c_int32(-1)
    if separate:
        ptrs = (flat.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(stride)).astype(np.uint64)
        _chk(L.kmx_debug_pack_strings(ptrs.ctypes.data, None, ln, ln, n, out.ctypes.data, C.byref(clean)))
    else:
        _chk(L.kmx_debug_pack_strings(None, flat.ctypes.data, ln, stride, n, out.ctypes.data, C.byref(clean)))
    return out, bool(clean.value)


def pack_strings_len(flat: np.ndarray, ln: int, separate: bool = False):
    """as pack_strings, for rows that hold `ln` characters followed by padding"""
    L = load_library()
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    return _pack(L, flat, flat.shape[0], ln, flat.shape[1], separate)


def kmc_list(db_prefix: str):
    """(k, total_kmers, kmers, counts) of a KMC database in listing order; host only."""
    L = load_library()
    k, total = C.c_int(0), C.c_uint64(0)
    _chk(L.kmx_kmc_info(db_prefix.encode(), C.byref(k), C.byref(total)))
    W = (k.value + 31) // 32
    km = np.zeros(max(total.value, 1) * W, dtype=np.uint64)
    cnt = np.zeros(max(total.value, 1), dtype=np.uint32)
    n = C.c_uint64(0)
    _chk(L.kmx_kmc_read(db_prefix.encode(), km.ctypes.data, cnt.ctypes.data, total.value, C.byref(n)))
    km = km[: n.value * W]
    return k.value, total.value, (km if W == 1 else km.reshape(-1, W)), cnt[: n.value]


def microbench(mode: int, nbytes: int, touches: int, iters: int = 3) -> float:
    s = C.c_double(0)
    _chk(load_library().kmx_microbench(mode, nbytes, touches, iters, C.byref(s)))
    return s.value


class KModel:
    """Handle on one model resident in HBM.  Mirrors reference class KModel (kmodel.hpp:39-672)."""

    DL = {"bf": 0, "bf_back": 1, "km_back": 2, "value": 3, "tag": 4, "claims": 5}

    def __init__(self, ci: int = 1, cs: int = 1023, num_hash: int = 7, num_bit: int = 5, _handle=None, device=None):
        """device: the HIP device of the handle (kmx_create_on); None = the calling thread's current device (kmx_create)"""
        self.L = load_library()
        if _handle is None:
            h = C.c_void_p()
            if device is None:
                _chk(self.L.kmx_create(ci, cs, num_hash, num_bit, C.byref(h)))
            else:
                _chk(self.L.kmx_create_on(int(device), ci, cs, num_hash, num_bit, C.byref(h)))
            _handle = h
        self.h = _handle

    # ---- build
    def init(self, db_file: str) -> None:                      # kmodel.hpp:57
        _chk(self.L.kmx_build_from_kmc(self.h, db_file.encode()))

    init_KModel = init                                         # README.md:76

    def build_packed(self, k: int, kmers: np.ndarray, counts: np.ndarray) -> None:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        _chk(self.L.kmx_build_host(self.h, k, kmers.ctypes.data, counts.ctypes.data, len(counts)))

    def build_dev(self, k: int, d_kmers_ptr: int, d_counts_ptr: int, n: int) -> None:
        _chk(self.L.kmx_build_dev(self.h, k, d_kmers_ptr, d_counts_ptr, n))

    def begin(self, k: int, n_bf, n_total: int) -> None:
        arr = (C.c_uint64 * 3)(*[int(x) for x in list(n_bf) + [0, 0, 0]][:3])
        _chk(self.L.kmx_begin(self.h, k, arr, n_total))

    def insert_batch(self, kmers: np.ndarray, counts: np.ndarray) -> None:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        _chk(self.L.kmx_insert_batch(self.h, kmers.ctypes.data, counts.ctypes.data, len(counts)))

    def insert_batch_dev(self, d_kmers_ptr: int, d_counts_ptr: int, n: int) -> None:
        _chk(self.L.kmx_insert_batch_dev(self.h, d_kmers_ptr, d_counts_ptr, n))

    def finish(self) -> None:
        _chk(self.L.kmx_finish(self.h))

    def set_stream(self, stream_ptr: int) -> None:
        _chk(self.L.kmx_set_stream(self.h, stream_ptr))

    # ---- one model, several GPUs: this rank's share (include/kmx.h, "ONE model built by several GPUs"; driven by dist.py)
    def count_classes_dev(self, d_counts_ptr: int, n: int):
        arr = (C.c_uint64 * 3)()
        _chk(self.L.kmx_count_classes_dev(self.h, d_counts_ptr, n, arr))
        return [int(x) for x in arr]

    def shard_begin(self, k: int, n_bf, n_total: int, rank: int, world: int) -> None:
        arr = (C.c_uint64 * 3)(*[int(x) for x in list(n_bf) + [0, 0, 0]][:3])
        _chk(self.L.kmx_shard_begin(self.h, k, arr, n_total, rank, world))

    def shard_classify_dev(self, d_kmers_ptr: int, d_counts_ptr: int, n: int, d_out_kmers_ptr: int, d_out_counts_ptr: int) -> int:
        n_out = C.c_uint64(0)
        _chk(self.L.kmx_shard_classify_dev(self.h, d_kmers_ptr, d_counts_ptr, n, d_out_kmers_ptr, d_out_counts_ptr, C.byref(n_out)))
        return int(n_out.value)

    def ring_msg_bytes(self, k: int) -> int:
        return int(self.L.kmx_ring_msg_bytes(k))

    def ring_round_dev(self, t: int, lists) -> None:
        """lists: (list, n_host, src_kmers_ptr, src_counts_ptr, src_msg_ptr, dst_msg_ptr) tuples, 0 for null pointers"""
        arr = (RingList * max(len(lists), 1))()
        for e, (i, n_host, sk, sc, sm, dm) in enumerate(lists):
            arr[e] = RingList(i, n_host, sk or None, sc or None, sm or None, dm or None)
        _chk(self.L.kmx_ring_round_dev(self.h, t, arr, len(lists)))

    def ring_stale_dup_dev(self, first_unused_row: int) -> None:
        _chk(self.L.kmx_ring_stale_dup_dev(self.h, first_unused_row))

    def shard_local(self):
        st, pk, pc = Stats(), C.c_void_p(), C.c_void_p()
        _chk(self.L.kmx_shard_local(self.h, C.byref(st), C.byref(pk), C.byref(pc)))
        return st, pk.value or 0, pc.value or 0

    def shard_complete(self, d_rest_kmers_ptr: int, d_rest_counts_ptr: int, n_rest: int, totals: Stats) -> None:
        _chk(self.L.kmx_shard_complete(self.h, d_rest_kmers_ptr or None, d_rest_counts_ptr or None, n_rest, C.byref(totals)))

    # ---- position-range partition (include/kmx.h, kmx_range_*)
    def range_begin(self, k: int, n_bf, n_total: int, rank: int, world: int) -> None:
        arr = (C.c_uint64 * 3)(*[int(x) for x in list(n_bf) + [0, 0, 0]][:3])
        _chk(self.L.kmx_range_begin(self.h, k, arr, n_total, rank, world))
        self._range_world = world

    def range_buffers(self):
        p, cap = C.c_void_p(), C.c_uint64()
        lo = (C.c_uint64 * (self._range_world + 1))()
        _chk(self.L.kmx_range_buffers(self.h, C.byref(p), C.byref(cap), lo))
        return p.value or 0, int(cap.value), [int(x) for x in lo]

    def range_emit_dev(self, t: int, lists):
        """-> (words per destination rank, commit words among them): the headers of the regions"""
        arr = (RingList * max(len(lists), 1))()
        for j, (i, n, pk, pc) in enumerate(lists):
            arr[j] = RingList(i, n, pk or None, pc or None, None, None)
        w = self._range_world
        counts = (C.c_uint64 * (2 * w))()
        _chk(self.L.kmx_range_emit_dev(self.h, t, arr, len(lists), counts))
        return [int(x) for x in counts[:w]], [int(x) for x in counts[w:]]

    def range_inband(self):
        """fixed-size messages: -> (device pointer of the regions, 64-bit words per region incl. its header, capx)"""
        p, rw, cx = C.c_void_p(), C.c_uint64(), C.c_uint64()
        _chk(self.L.kmx_range_inband(self.h, C.byref(p), C.byref(rw), C.byref(cx)))
        return p.value or 0, int(rw.value), int(cx.value)

    def range_emit_nowait_dev(self, t: int, lists) -> None:
        arr = (RingList * max(len(lists), 1))()
        for j, (i, n, pk, pc) in enumerate(lists):
            arr[j] = RingList(i, n, pk or None, pc or None, None, None)
        _chk(self.L.kmx_range_emit_dev(self.h, t, arr, len(lists), None))

    def range_flush_nowait_dev(self) -> None:
        _chk(self.L.kmx_range_flush_dev(self.h, None))

    def range_verdict_inband_dev(self, t: int, d_recv_ptr: int, n_src: int, d_verdict_ptr: int) -> None:
        _chk(self.L.kmx_range_verdict_inband_dev(self.h, t, d_recv_ptr, n_src, d_verdict_ptr))

    def range_commit_inband_dev(self, d_recv_ptr: int, n_src: int) -> None:
        _chk(self.L.kmx_range_commit_inband_dev(self.h, d_recv_ptr, n_src))

    def range_verdict_dev(self, t: int, d_words_ptr: int, totals, commits, d_verdict_ptr: int) -> None:
        n = len(totals)
        _chk(self.L.kmx_range_verdict_dev(self.h, t, d_words_ptr or None, (C.c_uint64 * n)(*totals), (C.c_uint64 * n)(*commits), n, d_verdict_ptr or None))

    def range_resolve_dev(self, t: int, d_verdict_ptr: int) -> None:
        _chk(self.L.kmx_range_resolve_dev(self.h, t, d_verdict_ptr or None))

    def range_commit_dev(self, d_commits_ptr: int, n: int) -> None:
        _chk(self.L.kmx_range_commit_dev(self.h, d_commits_ptr, n))

    def range_flush_dev(self):
        w = self._range_world
        counts = (C.c_uint64 * (2 * w))()
        _chk(self.L.kmx_range_flush_dev(self.h, counts))
        return [int(x) for x in counts[:w]]

    def dev_view(self, which: str, index: int = 0):
        """(device pointer, bytes) of a filter ("bf", "bf_back", "km_back") or of the cells of coupled array `index` ("cells")"""
        sel = {"bf": 0, "bf_back": 1, "km_back": 2, "cells": 3}[which]
        p, n = C.c_void_p(), C.c_uint64(0)
        _chk(self.L.kmx_dev_view(self.h, sel, index, C.byref(p), C.byref(n)))
        return p.value or 0, int(n.value)

    def or_words_dev(self, d_dst_ptr: int, d_src_ptr: int, n_words: int) -> None:
        _chk(self.L.kmx_or_words_dev(self.h, d_dst_ptr, d_src_ptr, n_words))

    # ---- query
    def kmer_to_occ(self, kmers, t_num: int = 4):             # kmodel.hpp:90,100 (t_num kept for signature parity)
        single = isinstance(kmers, str)
        strs = [kmers] if single else list(kmers)
        if not strs:
            return []
        out = np.zeros(len(strs), dtype=np.int32)
        by_len = {}
        for i, s in enumerate(strs):                           # the reference answers every string on its own
            by_len.setdefault(len(s), []).append(i)
        for ln, idx in by_len.items():
            part = np.zeros(len(idx), dtype=np.int32)
            _chk(self.L.kmx_query_ascii(self.h, "".join(strs[i] for i in idx).encode("latin-1"), ln, ln, len(idx), part.ctypes.data))
            out[idx] = part
        return int(out[0]) if single else out.tolist()

    def kmer_to_occ_rows(self, rows: np.ndarray, ln: int, separate: bool = True) -> np.ndarray:
        """kmer_to_occ over uint8[n, stride >= ln] rows holding one string each -- as n separate strings (kmx_query_strings:
        what the reference's vector<string> is, kmodel.hpp:90-98) or as one buffer (kmx_query_ascii)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n, stride = rows.shape
        out = np.zeros(n, dtype=np.int32)
        if separate:
            ptrs = (rows.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(stride)).astype(np.uint64)
            _chk(self.L.kmx_query_strings(self.h, C.cast(ptrs.ctypes.data, C.POINTER(C.c_char_p)), ln, n, out.ctypes.data))
        else:
            _chk(self.L.kmx_query_ascii(self.h, C.cast(rows.ctypes.data, C.c_char_p), ln, stride, n, out.ctypes.data))
        return out

    def kmer_to_occ_packed(self, kmers: np.ndarray) -> np.ndarray:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        k = self.stats().k
        if k == 0:
            raise KmxError(-4, "query before the model is built or loaded")
        n = kmers.size // ((k + 31) // 32)
        out = np.zeros(n, dtype=np.int32)
        _chk(self.L.kmx_query_packed(self.h, kmers.ctypes.data, n, out.ctypes.data))
        return out

    def kmer_to_occ_dev(self, d_kmers_ptr: int, n: int, d_out_ptr: int) -> None:
        _chk(self.L.kmx_query_packed_dev(self.h, d_kmers_ptr, n, d_out_ptr))

    # ---- persistence
    def save(self, save_dir: str) -> None:                    # kmodel.hpp:173
        _chk(self.L.kmx_save(self.h, save_dir.encode()))

    save_model = save                                          # README.md:78

    @classmethod
    def load(cls, save_dir: str) -> "KModel":                  # kmodel.hpp:680-696 + :209
        L = load_library()
        h = C.c_void_p()
        _chk(L.kmx_load(save_dir.encode(), C.byref(h)))
        return cls(_handle=h)

    load_model = load

    # ---- introspection
    def stats(self) -> Stats:
        st = Stats()
        _chk(self.L.kmx_get_stats(self.h, C.byref(st)))
        return st

    def download(self, which: str, index: int = 0) -> np.ndarray:
        st = self.stats()
        cap = max(int(st.km_byte_size), int(st.byte_km_back), max(st.byte_bf), max(st.byte_bf_back), 1)
        buf = np.zeros(cap, dtype=np.uint8)
        w = C.c_uint64(0)
        _chk(self.L.kmx_download(self.h, self.DL[which], index, buf.ctypes.data, cap, C.byref(w)))
        return buf[:w.value].copy()

    KERNEL_CLASSES = ["classify", "check", "commit", "slow_path", "reorder", "rest_table", "query", "detect", "commit_check", "file"]

    def set_profile(self, on) -> None:
        """True / 1: time the kernel classes; 2: the next builds run the fused launches' accounting variant (stats().piped_*)"""
        _chk(self.L.kmx_set_profile(self.h, int(on)))

    def kernel_times(self, reset: bool = True) -> dict:
        sec = (C.c_double * len(self.KERNEL_CLASSES))()
        cnt = (C.c_uint64 * len(self.KERNEL_CLASSES))()
        _chk(self.L.kmx_get_kernel_times(self.h, sec, cnt, int(reset)))
        return {n: {"seconds": sec[i], "launches": int(cnt[i])} for i, n in enumerate(self.KERNEL_CLASSES)}

    def build_seconds(self) -> float:
        a, b = C.c_double(0), C.c_double(0)
        _chk(self.L.kmx_last_build_seconds(self.h, C.byref(a), C.byref(b)))
        return b.value

    def close(self) -> None:
        if getattr(self, "h", None):
            self.L.kmx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


PARTITIONS = {"ring": 0, "range": 1, "range-rccl": 2}                                # KMX_PARTITION_* of include/kmx.h


def init_multi(models, db_file: str, partition: str = "ring") -> None:
    """KModel::init(db_file) (kmodel.hpp:57-86) by several handles together, from inside libkmx.so (kmx_build_from_kmc_multi_ex:
    one host thread per handle).  partition "ring": arrays owned whole, hipMemcpyPeerAsync hand-offs; "range": every array cut
    by position range, the words of a round written into the owners' inboxes through peer mappings, no host wait in a round.
    Every handle ends with the whole model."""
    L = load_library()
    arr = (C.c_void_p * len(models))(*[m.h for m in models])
    _chk(L.kmx_build_from_kmc_multi_ex(arr, len(models), db_file.encode(), PARTITIONS[partition]))


def get_model(ci_or_dir=1, cs: int = 1023, num_hash: int = 7, num_bit: int = 5) -> KModel:
    """Both reference factories (kmodel.hpp:674-677 and :680-696)."""
    if isinstance(ci_or_dir, (str, os.PathLike)):
        return KModel.load(os.fspath(ci_or_dir))
    return KModel(int(ci_or_dir), cs, num_hash, num_bit)
