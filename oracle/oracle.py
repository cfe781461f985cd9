"""ctypes front-end of the CPU oracle (oracle/caps_sa_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg may import this module, and only as the checker or
the reported CPU baseline.  Nothing in caps-sa_amd/ imports it.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcaps_sa_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (see oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("caps_sa_oracle.c", "caps_sa_oracle_impl.inc")]
    stale = (not os.path.exists(_LIB_PATH)
             or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcaps_sa_oracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u64, vp, ci = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int
        L.caps_oracle_lcp.restype = u64
        L.caps_oracle_lcp.argtypes = [vp, vp, u64]
        L.caps_oracle_remap.restype = None
        L.caps_oracle_remap.argtypes = [vp, u64]
        L.caps_oracle_max_threads.restype = ci
        for sfx in ("u32", "u64"):
            f = getattr(L, f"caps_oracle_build_{sfx}")
            f.restype = ci
            f.argtypes = [vp, u64, u64, u64, vp, vp, ci, vp]
            f = getattr(L, f"caps_oracle_merge_{sfx}")
            f.restype = None
            f.argtypes = [vp, u64, vp, u64, vp, u64, vp, vp, vp, vp]
            f = getattr(L, f"caps_oracle_merge_sort_{sfx}")
            f.restype = ci
            f.argtypes = [vp, u64, vp, u64, vp, vp]
            f = getattr(L, f"caps_oracle_upper_bound_{sfx}")
            f.restype = u64
            f.argtypes = [vp, u64, vp, u64, u64]
            f = getattr(L, f"caps_oracle_check_{sfx}")
            f.restype = ci
            f.argtypes = [vp, u64, vp, vp]
            f = getattr(L, f"caps_oracle_naive_{sfx}")
            f.restype = ci
            f.argtypes = [vp, u64, vp, vp]
        _lib = L
    return _lib


def _text(T) -> np.ndarray:
    if isinstance(T, (bytes, bytearray)):
        T = np.frombuffer(bytes(T), dtype=np.uint8)
    T = np.ascontiguousarray(T, dtype=np.uint8)
    return T


def _dt(idx_bits: int):
    return {32: (np.uint32, "u32"), 64: (np.uint64, "u64")}[idx_bits]


PHASES = ("sort_subarrays", "select_pivots", "locate_pivots", "collate", "merge_partitions",
          "boundary_lcp", "total")


def build_sa_lcp(T, p: int = 0, max_context: int = 0, idx_bits: int = 32, threads: int = 0,
                 timings: dict | None = None):
    """Reference algorithm (construct(), src/Suffix_Array.cpp:466-494) -> (SA, LCP)."""
    T = _text(T)
    n = T.size
    dt, sfx = _dt(idx_bits)
    SA = np.empty(n, dtype=dt)
    LCP = np.empty(n, dtype=dt)
    secs = (ctypes.c_double * 8)()
    if threads <= 0:
        threads = min(lib().caps_oracle_max_threads(), os.cpu_count() or 1)
    rc = getattr(lib(), f"caps_oracle_build_{sfx}")(T.ctypes.data, n, p, max_context,
                                                    SA.ctypes.data, LCP.ctypes.data, threads, secs)
    if rc == -1:
        raise ValueError("outside the reference's domain (needs n >= 32 and effective p >= 2)")
    if rc != 0:
        raise RuntimeError(f"caps_oracle_build_{sfx} failed: {rc}")
    if timings is not None:
        timings.update({k: secs[i] for i, k in enumerate(PHASES)})
        timings["threads"] = threads
    return SA, LCP


def naive_sa_lcp(T, idx_bits: int = 32):
    """Independent comparison-sort construction (cross-check only)."""
    T = _text(T)
    dt, sfx = _dt(idx_bits)
    SA = np.empty(T.size, dtype=dt)
    LCP = np.empty(T.size, dtype=dt)
    getattr(lib(), f"caps_oracle_naive_{sfx}")(T.ctypes.data, T.size, SA.ctypes.data, LCP.ctypes.data)
    return SA, LCP


def check(T, SA, LCP) -> int:
    """0 iff (SA, LCP) is exactly the suffix array / LCP array of T."""
    T = _text(T)
    idx_bits = 32 if SA.dtype == np.uint32 else 64
    dt, sfx = _dt(idx_bits)
    SA = np.ascontiguousarray(SA, dtype=dt)
    LCP = np.ascontiguousarray(LCP, dtype=dt)
    return getattr(lib(), f"caps_oracle_check_{sfx}")(T.ctypes.data, T.size, SA.ctypes.data, LCP.ctypes.data)


def lcp(T, a: int, b: int) -> int:
    T = _text(T)
    n = T.size
    return lib().caps_oracle_lcp(T.ctypes.data + a, T.ctypes.data + b, n - max(a, b))


def merge(T, X, Y, LX, LY, idx_bits: int = 32):
    T = _text(T)
    dt, sfx = _dt(idx_bits)
    X, Y, LX, LY = (np.ascontiguousarray(v, dtype=dt) for v in (X, Y, LX, LY))
    Z = np.empty(X.size + Y.size, dtype=dt)
    LZ = np.empty_like(Z)
    getattr(lib(), f"caps_oracle_merge_{sfx}")(T.ctypes.data, T.size, X.ctypes.data, X.size, Y.ctypes.data,
                                               Y.size, LX.ctypes.data, LY.ctypes.data, Z.ctypes.data,
                                               LZ.ctypes.data)
    return Z, LZ


def merge_sort(T, idxs, idx_bits: int = 32):
    T = _text(T)
    dt, sfx = _dt(idx_bits)
    idxs = np.ascontiguousarray(idxs, dtype=dt)
    Y = np.empty_like(idxs)
    L = np.empty_like(idxs)
    rc = getattr(lib(), f"caps_oracle_merge_sort_{sfx}")(T.ctypes.data, T.size, idxs.ctypes.data, idxs.size,
                                                         Y.ctypes.data, L.ctypes.data)
    assert rc == 0
    return Y, L


def upper_bound(T, X, pivot: int, idx_bits: int = 32) -> int:
    T = _text(T)
    dt, sfx = _dt(idx_bits)
    X = np.ascontiguousarray(X, dtype=dt)
    return getattr(lib(), f"caps_oracle_upper_bound_{sfx}")(T.ctypes.data, T.size, X.ctypes.data, X.size, pivot)


def remap(raw) -> np.ndarray:
    """CLI byte remap, src/main.cpp:61-70."""
    T = _text(raw).copy()
    lib().caps_oracle_remap(T.ctypes.data, T.size)
    return T


def dump_bytes(SA: np.ndarray, LCP: np.ndarray) -> bytes:
    """On-disk format of Suffix_Array::dump, src/Suffix_Array.cpp:497-509."""
    return struct.pack("<Q", SA.size) + SA.tobytes() + LCP.tobytes()


def dump_sha256(SA: np.ndarray, LCP: np.ndarray) -> str:
    h = hashlib.sha256()
    h.update(struct.pack("<Q", SA.size))
    h.update(memoryview(np.ascontiguousarray(SA)).cast("B"))
    h.update(memoryview(np.ascontiguousarray(LCP)).cast("B"))
    return h.hexdigest()


def gen_rand_seq(seed: int, N: int, newline: bool = True) -> np.ndarray:
    """Bit-identical output of utils/gen_rand_seq.py (reference utils/gen_rand_seq.py:9-16):
    random.seed(seed); N x random.choice('ACGT'); print() appends a newline.

    random.choice -> _randbelow(4) -> getrandbits(3) with rejection of values >= 4, and
    getrandbits(3) is the top 3 bits of one 32-bit MT19937 output.  The Mersenne Twister
    state of CPython's generator is transplanted into numpy's MT19937 for speed.
    """
    import random
    rs = random.Random(seed)
    st = rs.getstate()[1]
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": np.array(st[:-1], dtype=np.uint32), "pos": st[-1]}}
    out = np.empty(N + (1 if newline else 0), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    filled = 0
    while filled < N:
        want = N - filled
        raw = bg.random_raw(int(want * 2.1) + 64).astype(np.uint32) >> np.uint32(29)
        keep = raw[raw < 4]
        take = min(want, keep.size)
        if take < keep.size:
            # Unused draws would desynchronise a later call; this function is one-shot, so
            # dropping them is harmless.
            keep = keep[:take]
        out[filled:filled + take] = lut[keep]
        filled += take
    if newline:
        out[N] = ord("\n")
    return out
