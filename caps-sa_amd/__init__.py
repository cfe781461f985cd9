"""caps-sa_amd: MI355X-native suffix-array / LCP-array construction (SA+LCP of a text).

Host-side mirror of the reference's class surface ``CaPS_SA::Suffix_Array<idx>``
(reference include/Suffix_Array.hpp:148-181) for Python callers; the C++ mirror is
caps-sa_amd/csrc/Suffix_Array.hpp.  All work happens in libcaps_sa_hip.so (HIP kernels for
gfx950) through the C ABI of include/caps_sa_hip.h.  There is no CPU fallback: importing
works without a GPU (so that the build can be checked), but any computation without the
HIP library or without a device raises.
"""
from __future__ import annotations

import os
import subprocess

import numpy as np

from ._binding import CapsLib, CapsSaError, Stats, Shard, ShardInfo, EXPORTS  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# CAPS_SA_LIB selects a tuning variant built by `make variant` (benchmarking only).
LIB_PATH = os.environ.get("CAPS_SA_LIB") or os.path.join(_HERE, "libcaps_sa_hip.so")
_lib: CapsLib | None = None


def build_library(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "caps_sa_hip.h"))
    stale = (not os.path.exists(LIB_PATH)
             or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcaps_sa_hip.so"])
    return LIB_PATH


def lib() -> CapsLib:
    """The bound product library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        _lib = CapsLib(LIB_PATH, "caps_sa_hip_")
    return _lib


class SuffixArray:
    """Python mirror of ``CaPS_SA::Suffix_Array<T_idx_>`` (include/Suffix_Array.hpp:22-181).

    ``SuffixArray(T, subproblem_count=0, max_context=0)``; ``construct()``; ``SA()``;
    ``LCP()``; ``T()``; ``n()``; ``dump(path)``.  Index width follows src/main.cpp:76-87
    unless ``idx_bits`` is given.
    """

    def __init__(self, T, subproblem_count: int = 0, max_context: int = 0, idx_bits: int | None = None, device: int = 0,
                 devices: list[int] | None = None, pinned: bool = False):
        self._T = CapsLib._text(T)
        self._n = int(self._T.size)
        self._p = int(subproblem_count)
        self._ctx = int(max_context)
        self._bits = idx_bits or (32 if self._n <= 0xFFFFFFFF else 64)
        self._device = device
        self._devices = list(devices) if devices else None      # several GPUs from this process (caps_sa_hip_build_multi_*)
        self._pinned = pinned                                    # page-locked result arrays (what the C++ mirror allocates)
        self._SA = None
        self._LCP = None
        self.stats: dict | None = None

    def T(self) -> np.ndarray:
        return self._T

    def n(self) -> int:
        return self._n

    def construct(self) -> None:
        if self._devices:
            self._SA, self._LCP, self.stats = lib().build_multi(self._T, self._devices, self._p, self._ctx, self._bits, self._pinned)
        else:
            self._SA, self._LCP, self.stats = lib().build(self._T, self._p, self._ctx, self._bits, self._device, self._pinned)

    def SA(self) -> np.ndarray:
        if self._SA is None:
            raise RuntimeError("construct() has not been called")
        return self._SA

    def LCP(self) -> np.ndarray:
        if self._LCP is None:
            raise RuntimeError("construct() has not been called")
        return self._LCP

    def dump(self, path: str) -> None:
        """Suffix_Array::dump format (src/Suffix_Array.cpp:497-509): u64 n, SA, LCP."""
        with open(path, "wb") as f:
            f.write(np.uint64(self._n).tobytes())
            f.write(self.SA().tobytes())
            f.write(self.LCP().tobytes())
