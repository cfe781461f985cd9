"""ctypes binding of the C ABI declared in include/caps_sa_hip.h.

``CapsLib(path, prefix)`` binds one shared library.  The product package binds
libcaps_sa_hip.so (prefix ``caps_sa_hip_``); tests bind the host emulation of the same
sources (tests/emul/libcaps_sa_emul.so, prefix ``caps_sa_emul_``) through this class too.
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np

_u64, _vp, _ci = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int


class Stats(ctypes.Structure):
    """caps_sa_stats (include/caps_sa_hip.h)."""
    _fields_ = [
        ("n", ctypes.c_uint64),
        ("idx_bytes", ctypes.c_uint32),
        ("p_eff", ctypes.c_uint32),
        ("ppp", ctypes.c_uint32),
        ("bits_per_char", ctypes.c_uint32),
        ("merge_passes_phase1", ctypes.c_uint32),
        ("merge_passes_phase2", ctypes.c_uint32),
        ("merge_passes_samples", ctypes.c_uint32),
        ("long_runs", ctypes.c_uint32),
        ("max_partition", ctypes.c_uint64),
        ("workspace_bytes", ctypes.c_uint64),
        ("ms_total", ctypes.c_double),
        ("ms_pack", ctypes.c_double),
        ("ms_sort_subarrays", ctypes.c_double),
        ("ms_select_pivots", ctypes.c_double),
        ("ms_locate_pivots", ctypes.c_double),
        ("ms_partition", ctypes.c_double),
        ("ms_merge_partitions", ctypes.c_double),
        ("ms_boundary_lcp", ctypes.c_double),
        ("ms_output", ctypes.c_double),
        ("ms_h2d", ctypes.c_double),
        ("ms_d2h", ctypes.c_double),
        ("merge_pass_ms", ctypes.c_double),
        ("merge_pass_launches", ctypes.c_uint64),
        ("merge_pass_elems", ctypes.c_uint64),
        ("tile_sort_ms", ctypes.c_double),
        ("tile_sort_launches", ctypes.c_uint64),
        ("tile_sort_elems", ctypes.c_uint64),
        ("bucket_scatter_ms", ctypes.c_double),
        ("bucket_scatter_launches", ctypes.c_uint64),
        ("bucket_scatter_elems", ctypes.c_uint64),
        ("bucket_count_ms", ctypes.c_double),
        ("collate_ms", ctypes.c_double),
        ("slot_splits", ctypes.c_uint32),
        ("slot_splits_redone", ctypes.c_uint32),
        ("path_direct", ctypes.c_uint32),
        ("path_fallback", ctypes.c_uint32),
        ("direct_groups", ctypes.c_uint32),
        ("direct_quantile", ctypes.c_uint32),
        ("direct_max_group", ctypes.c_uint64),
        ("level_a_ms", ctypes.c_double),
        ("direct_key_bits", ctypes.c_uint32),
        ("run_buckets", ctypes.c_uint32),
        ("result_waves", ctypes.c_uint32),
        ("n_devices", ctypes.c_uint32),
        ("ms_upload_max", ctypes.c_double),
        ("ms_upload_min", ctypes.c_double),
        ("ms_device_build_max", ctypes.c_double),
        ("ms_device_build_min", ctypes.c_double),
        ("ms_download_max", ctypes.c_double),
        ("ms_download_min", ctypes.c_double),
        ("tie_groups_deferred", ctypes.c_uint64),
        ("tie_elems_deferred", ctypes.c_uint64),
        ("tie_levels", ctypes.c_uint32),
        ("lcp_bytes_on_link", ctypes.c_uint32),
        ("finish_ms", ctypes.c_double),
        ("run_bucket_ms", ctypes.c_double),
        ("msd_ms", ctypes.c_double),
        ("knot_slot_splits", ctypes.c_uint32),
        ("knot_slot_splits_redone", ctypes.c_uint32),
        ("spill_entries", ctypes.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


class ShardInfo(ctypes.Structure):
    """caps_sa_shard_info (include/caps_sa_hip.h)."""
    _fields_ = [
        ("n", ctypes.c_uint64),
        ("p", ctypes.c_uint32), ("ppp", ctypes.c_uint32), ("rank", ctypes.c_uint32), ("world", ctypes.c_uint32),
        ("g0", ctypes.c_uint32), ("g1", ctypes.c_uint32),
        ("bits_per_char", ctypes.c_uint32), ("idx_bytes", ctypes.c_uint32),
        ("part_lo", ctypes.c_uint32), ("part_hi", ctypes.c_uint32),
        ("local_elems", ctypes.c_uint64),
        ("m_local", ctypes.c_uint64), ("m_total", ctypes.c_uint64),
        ("recv_total", ctypes.c_uint64),
        ("slice_off", ctypes.c_uint64),
        ("capacity", ctypes.c_uint64),
        ("ms_phase1", ctypes.c_double), ("ms_pivots", ctypes.c_double), ("ms_collate", ctypes.c_double),
        ("ms_phase2", ctypes.c_double),
        ("direct_fallback", ctypes.c_uint32), ("direct_groups", ctypes.c_uint32), ("direct_sub", ctypes.c_uint32),
        ("n_streams", ctypes.c_uint32),
        ("stream_cap", ctypes.c_uint64), ("send_capacity", ctypes.c_uint64),
        ("ms_scatter", ctypes.c_double), ("ms_sort", ctypes.c_double),
        ("ms_level_a", ctypes.c_double), ("ms_level_b", ctypes.c_double), ("ms_tile_sort", ctypes.c_double),
        ("ms_merge_passes", ctypes.c_double),
        ("level_a_elems", ctypes.c_uint64),
        ("slot_splits", ctypes.c_uint32), ("slot_splits_redone", ctypes.c_uint32),
        ("key_bytes", ctypes.c_uint32), ("exchange", ctypes.c_uint32),
        ("direct_quantile", ctypes.c_uint32), ("run_buckets", ctypes.c_uint32),
        ("tie_groups_deferred", ctypes.c_uint64), ("tie_levels", ctypes.c_uint32), ("reserved_", ctypes.c_uint32),
    ]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


SHARD_EXPORTS = ["shard_create", "shard_destroy", "shard_info", "shard_phase1", "shard_pivots", "shard_collate",
                 "shard_phase2", "shard_last_sa", "shard_fix_first_lcp", "shard_scatter", "shard_plan", "shard_sort",
                 "shard_phase1_arrays", "shard_set_key_bits"]

EXPORTS = ["device_count", "last_error", "version", "stats_bytes", "shard_info_bytes", "workspace_bytes", "workspace_bytes_ex", "release_cache", "host_alloc", "host_free", "gen_rand_seq"] + SHARD_EXPORTS + [
    f"{name}_{sfx}"
    for sfx in ("u32", "u64")
    for name in ("build", "build_multi", "build_device", "verify_device", "verify_slice_device", "sort_suffixes", "sort_segments", "merge",
                 "upper_bound", "lcp")
]


class CapsSaError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"caps_sa error {code}: {msg}")
        self.code = code


def _sfx(idx_bits: int):
    return {32: ("u32", np.uint32), 64: ("u64", np.uint64)}[idx_bits]


class CapsLib:
    def __init__(self, path: str, prefix: str = "caps_sa_hip_"):
        self.path = path
        self.prefix = prefix
        self.dll = ctypes.CDLL(path)
        f = self._f
        f("device_count").restype = _ci
        f("device_count").argtypes = []
        f("last_error").restype = ctypes.c_char_p
        f("version").restype = ctypes.c_char_p
        # the structs grow at their end from release to release and the library writes all of them: a mirror of another size
        # would be overrun (or read short) -- refuse to work with it
        for name, mirror in (("stats_bytes", Stats), ("shard_info_bytes", ShardInfo)):
            f(name).restype = ctypes.c_uint32
            f(name).argtypes = []
            if f(name)() != ctypes.sizeof(mirror):
                raise CapsSaError(-1, f"{path}: its {name} = {f(name)()}, this binding's mirror has {ctypes.sizeof(mirror)} "
                                      "(library and _binding.py are of different versions)")
        f("workspace_bytes").restype = _ci
        f("workspace_bytes").argtypes = [_u64, _u64, _ci, ctypes.POINTER(_u64)]
        f("workspace_bytes_ex").restype = _ci
        f("workspace_bytes_ex").argtypes = [_u64, _u64, _ci, _ci, ctypes.POINTER(_u64)]
        f("release_cache").restype = None
        f("release_cache").argtypes = []
        f("host_alloc").restype = _vp
        f("host_alloc").argtypes = [_u64]
        f("host_free").restype = None
        f("host_free").argtypes = [_vp]
        f("gen_rand_seq").restype = _ci
        f("gen_rand_seq").argtypes = [ctypes.c_uint32, _u64, _vp]
        for sfx in ("u32", "u64"):
            f(f"build_{sfx}").restype = _ci
            f(f"build_{sfx}").argtypes = [_vp, _u64, _u64, _u64, _vp, _vp, _ci, ctypes.POINTER(Stats)]
            f(f"build_multi_{sfx}").restype = _ci
            f(f"build_multi_{sfx}").argtypes = [_vp, _u64, _u64, _u64, _vp, _vp, _vp, _ci, ctypes.POINTER(Stats)]
            f(f"build_device_{sfx}").restype = _ci
            f(f"build_device_{sfx}").argtypes = [_vp, _u64, _u64, _u64, _vp, _vp, _vp, _u64, _vp, ctypes.POINTER(Stats)]
            f(f"verify_device_{sfx}").restype = _ci
            f(f"verify_device_{sfx}").argtypes = [_vp, _u64, _vp, _vp, _vp, ctypes.POINTER(_u64)]
            f(f"verify_slice_device_{sfx}").restype = _ci
            f(f"verify_slice_device_{sfx}").argtypes = [_vp, _u64, _vp, _vp, _u64, _ci, _vp, ctypes.POINTER(_u64)]
            f(f"sort_suffixes_{sfx}").restype = _ci
            f(f"sort_suffixes_{sfx}").argtypes = [_vp, _u64, _vp, _u64, _vp, _vp, _ci]
            f(f"sort_segments_{sfx}").restype = _ci
            f(f"sort_segments_{sfx}").argtypes = [_vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _ci]
            f(f"merge_{sfx}").restype = _ci
            f(f"merge_{sfx}").argtypes = [_vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _vp, _vp, _ci]
            f(f"upper_bound_{sfx}").restype = _ci
            f(f"upper_bound_{sfx}").argtypes = [_vp, _u64, _vp, _u64, _vp, _u64, _vp, _ci]
            f(f"lcp_{sfx}").restype = _ci
            f(f"lcp_{sfx}").argtypes = [_vp, _u64, _vp, _vp, _u64, _vp, _ci]

        f("shard_create").restype = _ci
        f("shard_create").argtypes = [_vp, _u64, _u64, _ci, _ci, _ci, _vp, ctypes.POINTER(_vp)]
        f("shard_destroy").restype = None
        f("shard_destroy").argtypes = [_vp]
        f("shard_info").restype = _ci
        f("shard_info").argtypes = [_vp, ctypes.POINTER(ShardInfo)]
        f("shard_phase1").restype = _ci
        f("shard_phase1").argtypes = [_vp, _vp, _vp]
        f("shard_pivots").restype = _ci
        f("shard_pivots").argtypes = [_vp, _vp, _vp, _vp]
        f("shard_collate").restype = _ci
        f("shard_collate").argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
        f("shard_phase2").restype = _ci
        f("shard_phase2").argtypes = [_vp, _vp, _vp, _vp, _vp]
        f("shard_scatter").restype = _ci
        f("shard_scatter").argtypes = [_vp, _vp, _vp, _vp]
        f("shard_plan").restype = _ci
        f("shard_plan").argtypes = [_vp, _vp, _vp, _vp]
        f("shard_sort").restype = _ci
        f("shard_sort").argtypes = [_vp, _vp, _vp, _vp, _vp]
        f("shard_set_key_bits").restype = _ci
        f("shard_set_key_bits").argtypes = [_vp, _ci]
        f("shard_phase1_arrays").restype = _ci
        f("shard_phase1_arrays").argtypes = [_vp, _vp, _vp, ctypes.POINTER(_u64), ctypes.POINTER(_u64)]
        f("shard_last_sa").restype = _ci
        f("shard_last_sa").argtypes = [_vp, ctypes.POINTER(_u64)]
        f("shard_fix_first_lcp").restype = _ci
        f("shard_fix_first_lcp").argtypes = [_vp, _u64, _vp]

    def _f(self, name: str):
        return getattr(self.dll, self.prefix + name)

    def shard(self, dT_ptr: int, n: int, p: int, idx_bits: int, rank: int, world: int, stream: int = 0) -> "Shard":
        return Shard(self, dT_ptr, n, p, idx_bits, rank, world, stream)

    def _check(self, rc: int):
        if rc != 0:
            raise CapsSaError(rc, self._f("last_error")().decode(errors="replace"))

    # ------------------------------------------------------------------ queries
    def device_count(self) -> int:
        return self._f("device_count")()

    def version(self) -> str:
        return self._f("version")().decode()

    def workspace_bytes(self, n: int, p: int = 0, idx_bits: int = 32, bits_per_char: int = 8) -> int:
        """bits_per_char = 2: a text of at most 4 distinct bytes (anything behind the reference CLI) -- a smaller text arena."""
        out = _u64(0)
        self._check(self._f("workspace_bytes_ex")(n, p, idx_bits // 8, bits_per_char, ctypes.byref(out)))
        return out.value

    # ------------------------------------------------------------------ host-buffer build
    @staticmethod
    def _text(T) -> np.ndarray:
        if isinstance(T, (bytes, bytearray)):
            T = np.frombuffer(bytes(T), dtype=np.uint8)
        return np.ascontiguousarray(T, dtype=np.uint8)

    def release_cache(self) -> None:
        """Frees the device memory the host-buffer entry points keep between calls."""
        self._f("release_cache")()

    def gen_rand_seq(self, seed: int, n: int, out: np.ndarray | None = None) -> np.ndarray:
        """n letters of the reference's utils/gen_rand_seq.py stream (host; see include/caps_sa_hip.h)."""
        if out is None:
            out = np.empty(n, dtype=np.uint8)
        assert out.dtype == np.uint8 and out.size >= n and out.flags.c_contiguous
        self._check(self._f("gen_rand_seq")(seed, n, out.ctypes.data))
        return out[:n]

    def pinned_empty(self, count: int, dtype) -> np.ndarray:
        """numpy array over page-locked host memory (caps_sa_hip_host_alloc); freed with the array."""
        dtype = np.dtype(dtype)
        nbytes = max(1, count * dtype.itemsize)
        ptr = self._f("host_alloc")(nbytes)
        if not ptr:
            raise CapsSaError(-12, (self._f("last_error")() or b"host_alloc failed").decode())
        buf = (ctypes.c_char * nbytes).from_address(ptr)
        arr = np.frombuffer(buf, dtype=dtype, count=count)
        free = self._f("host_free")
        weakref.finalize(buf, free, ptr)
        return arr

    def build(self, T, p: int = 0, max_context: int = 0, idx_bits: int = 32, device: int = 0, pinned: bool = False):
        """construct() on host buffers -> (SA, LCP, stats dict).  pinned: SA / LCP in page-locked memory."""
        T = self._text(T)
        sfx, dt = _sfx(idx_bits)
        SA = self.pinned_empty(T.size, dt) if pinned else np.empty(T.size, dtype=dt)
        LCP = self.pinned_empty(T.size, dt) if pinned else np.empty(T.size, dtype=dt)
        st = Stats()
        self._check(self._f(f"build_{sfx}")(T.ctypes.data, T.size, p, max_context, SA.ctypes.data, LCP.ctypes.data,
                                            device, ctypes.byref(st)))
        return SA, LCP, st.as_dict()

    def build_into(self, T, SA: np.ndarray, LCP: np.ndarray, p: int = 0, max_context: int = 0, idx_bits: int = 32, device: int = 0) -> dict:
        """construct() into caller-owned result arrays (e.g. pinned_empty ones, re-used between calls) -> stats dict."""
        T = self._text(T)
        sfx, dt = _sfx(idx_bits)
        assert SA.dtype == dt and LCP.dtype == dt and SA.size >= T.size and LCP.size >= T.size
        st = Stats()
        self._check(self._f(f"build_{sfx}")(T.ctypes.data, T.size, p, max_context, SA.ctypes.data, LCP.ctypes.data,
                                            device, ctypes.byref(st)))
        return st.as_dict()

    def build_multi(self, T, devices, p: int = 0, max_context: int = 0, idx_bits: int = 32, pinned: bool = False):
        """construct() on several GPUs from this process (caps_sa_hip_build_multi_*) -> (SA, LCP, stats dict)."""
        T = self._text(T)
        sfx, dt = _sfx(idx_bits)
        SA = self.pinned_empty(T.size, dt) if pinned else np.empty(T.size, dtype=dt)
        LCP = self.pinned_empty(T.size, dt) if pinned else np.empty(T.size, dtype=dt)
        devs = (ctypes.c_int * len(devices))(*devices)
        st = Stats()
        self._check(self._f(f"build_multi_{sfx}")(T.ctypes.data, T.size, p, max_context, SA.ctypes.data, LCP.ctypes.data,
                                                  devs, len(devices), ctypes.byref(st)))
        return SA, LCP, st.as_dict()

    # ------------------------------------------------------------------ device-resident build
    def build_device(self, dT_ptr: int, n: int, dSA_ptr: int, dLCP_ptr: int, p: int = 0, max_context: int = 0,
                     idx_bits: int = 32, workspace_ptr: int = 0, workspace_bytes: int = 0, stream: int = 0) -> dict:
        sfx, _ = _sfx(idx_bits)
        st = Stats()
        self._check(self._f(f"build_device_{sfx}")(dT_ptr, n, p, max_context, dSA_ptr, dLCP_ptr, workspace_ptr or None,
                                                   workspace_bytes, stream or None, ctypes.byref(st)))
        return st.as_dict()

    def verify_device(self, dT_ptr: int, n: int, dSA_ptr: int, dLCP_ptr: int, idx_bits: int = 32, stream: int = 0) -> int:
        sfx, _ = _sfx(idx_bits)
        err = _u64(0)
        self._check(self._f(f"verify_device_{sfx}")(dT_ptr, n, dSA_ptr, dLCP_ptr, stream or None, ctypes.byref(err)))
        return err.value

    def verify_slice_device(self, dT_ptr: int, n: int, dSA_ptr: int, dLCP_ptr: int, cnt: int, is_head: bool, idx_bits: int = 32,
                            stream: int = 0) -> int:
        sfx, _ = _sfx(idx_bits)
        err = _u64(0)
        self._check(self._f(f"verify_slice_device_{sfx}")(dT_ptr, n, dSA_ptr, dLCP_ptr, cnt, 1 if is_head else 0, stream or None,
                                                          ctypes.byref(err)))
        return err.value

    # ------------------------------------------------------------------ kernel-level entry points
    def sort_suffixes(self, T, idx, idx_bits: int = 32, device: int = 0):
        T = self._text(T)
        sfx, dt = _sfx(idx_bits)
        idx = np.ascontiguousarray(idx, dtype=dt)
        out_sa = np.empty_like(idx)
        out_lcp = np.empty_like(idx)
        self._check(self._f(f"sort_suffixes_{sfx}")(T.ctypes.data, T.size, idx.ctypes.data, idx.size,
                                                    out_sa.ctypes.data, out_lcp.ctypes.data, device))
        return out_sa, out_lcp

    def sort_segments(self, T, idx, seg_start, idx_bits: int = 32, device: int = 0):
        T = self._text(T)
        sfx, dt = _sfx(idx_bits)
        idx = np.ascontiguousarray(idx, dtype=dt)
        seg = np.ascontiguousarray(seg_start, dtype=np.uint64)
        out_sa = np.empty_like(idx)
        out_lcp = np.empty_like(idx)
        self._check(self._f(f"sort_segments_{sfx}")(T.ctypes.data, T.size, idx.ctypes.data, idx.size, seg.ctypes.data,
                                                    seg.size - 1, out_sa.ctypes.data, out_lcp.ctypes.data, device))
        return out_sa, out_lcp

    def merge(self, T, X, Y, LX, LY, idx_bits: int = 32, device: int = 0):
        T = self._text(T)
        sfx, dt = _sfx(idx_bits)
        X, Y, LX, LY = (np.ascontiguousarray(v, dtype=dt) for v in (X, Y, LX, LY))
        Z = np.empty(X.size + Y.size, dtype=dt)
        LZ = np.empty_like(Z)
        self._check(self._f(f"merge_{sfx}")(T.ctypes.data, T.size, X.ctypes.data, X.size, Y.ctypes.data, Y.size,
                                            LX.ctypes.data, LY.ctypes.data, Z.ctypes.data, LZ.ctypes.data, device))
        return Z, LZ

    def upper_bound(self, T, X, pivots, idx_bits: int = 32, device: int = 0):
        T = self._text(T)
        sfx, dt = _sfx(idx_bits)
        X = np.ascontiguousarray(X, dtype=dt)
        pivots = np.ascontiguousarray(pivots, dtype=dt)
        out = np.empty_like(pivots)
        self._check(self._f(f"upper_bound_{sfx}")(T.ctypes.data, T.size, X.ctypes.data, X.size, pivots.ctypes.data,
                                                  pivots.size, out.ctypes.data, device))
        return out

    def lcp(self, T, a, b, idx_bits: int = 32, device: int = 0):
        T = self._text(T)
        sfx, dt = _sfx(idx_bits)
        a = np.ascontiguousarray(a, dtype=dt)
        b = np.ascontiguousarray(b, dtype=dt)
        out = np.empty_like(a)
        self._check(self._f(f"lcp_{sfx}")(T.ctypes.data, T.size, a.ctypes.data, b.ctypes.data, a.size,
                                          out.ctypes.data, device))
        return out


class Shard:
    """One rank of the multi-GPU construction (caps_sa_hip_shard_*, include/caps_sa_hip.h).
    Pointers are raw device addresses (torch tensors' data_ptr())."""

    def __init__(self, lib: CapsLib, dT_ptr: int, n: int, p: int, idx_bits: int, rank: int, world: int, stream: int = 0):
        self.lib = lib
        self.h = _vp(None)
        lib._check(lib._f("shard_create")(dT_ptr, n, p, idx_bits // 8, rank, world, stream or None, ctypes.byref(self.h)))

    def close(self):
        if self.h:
            self.lib._f("shard_destroy")(self.h)
            self.h = _vp(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        inf = ShardInfo()
        self.lib._check(self.lib._f("shard_info")(self.h, ctypes.byref(inf)))
        return inf.as_dict()

    def phase1(self, d_sample_keys: int, d_sample_sa: int):
        self.lib._check(self.lib._f("shard_phase1")(self.h, d_sample_keys or None, d_sample_sa or None))

    def pivots(self, d_all_keys: int, d_all_sa: int, d_local_sizes: int):
        self.lib._check(self.lib._f("shard_pivots")(self.h, d_all_keys, d_all_sa, d_local_sizes))

    def collate(self, all_sizes: np.ndarray, d_send_keys: int, d_send_sa: int):
        all_sizes = np.ascontiguousarray(all_sizes, dtype=np.uint64)
        world = all_sizes.shape[0]
        sc = np.zeros(world, dtype=np.uint64)
        rc = np.zeros(world, dtype=np.uint64)
        self.lib._check(self.lib._f("shard_collate")(self.h, all_sizes.ctypes.data, d_send_keys or None, d_send_sa or None,
                                                     sc.ctypes.data, rc.ctypes.data))
        return sc, rc

    def phase2(self, d_recv_keys: int, d_recv_sa: int, dSA: int, dLCP: int):
        self.lib._check(self.lib._f("shard_phase2")(self.h, d_recv_keys or None, d_recv_sa or None, dSA or None, dLCP or None))

    def scatter(self, d_send_keys: int, d_send_sa: int, d_report: int):
        self.lib._check(self.lib._f("shard_scatter")(self.h, d_send_keys, d_send_sa, d_report))

    def plan(self, all_reports: np.ndarray):
        """-> (fallback code, send_counts, recv_counts); code 0: go on with the direct path."""
        all_reports = np.ascontiguousarray(all_reports, dtype=np.uint64)
        world = all_reports.shape[0]
        sc = np.zeros(world, dtype=np.uint64)
        rc = np.zeros(world, dtype=np.uint64)
        code = self.lib._f("shard_plan")(self.h, all_reports.ctypes.data, sc.ctypes.data, rc.ctypes.data)
        if code < 0:
            self.lib._check(code)
        return code, sc, rc

    def sort_owned(self, d_recv_keys: int, d_recv_sa: int, dSA: int, dLCP: int) -> int:
        """-> 0, or CAPS_SA_FB_KEY32 (6): a slot overflowed under 32-bit keys; all ranks set_key_bits(64) and start over."""
        code = self.lib._f("shard_sort")(self.h, d_recv_keys or None, d_recv_sa or None, dSA or None, dLCP or None)
        if code < 0:
            self.lib._check(code)
        return code

    def set_key_bits(self, bits: int):
        self.lib._check(self.lib._f("shard_set_key_bits")(self.h, bits))

    def phase1_arrays(self, d_keys_out: int = 0, d_sa_out: int = 0):
        """Copies the rank's sorted subarrays (after phase1()) into the given device buffers; -> (count, subarray length)."""
        c, sl = _u64(0), _u64(0)
        self.lib._check(self.lib._f("shard_phase1_arrays")(self.h, d_keys_out or None, d_sa_out or None, ctypes.byref(c), ctypes.byref(sl)))
        return c.value, sl.value

    def last_sa(self) -> int:
        v = _u64(0)
        self.lib._check(self.lib._f("shard_last_sa")(self.h, ctypes.byref(v)))
        return v.value

    def fix_first_lcp(self, prev_sa: int, dLCP: int):
        self.lib._check(self.lib._f("shard_fix_first_lcp")(self.h, prev_sa, dLCP or None))
