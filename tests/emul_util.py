"""Loader of the host emulation of the HIP kernels (tests/emul, test infrastructure)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "emul")
_emul = None


def emul():
    global _emul
    if os.environ.get("CAPS_EMUL_REV") == "1":      # the whole suite through the reversed-order builds (a one-off check)
        return emul_rev(False)
    if _emul is None:
        subprocess.check_call(["make", "-s", "-C", EMUL_DIR, "libcaps_sa_emul.so"])
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import caps_sa_amd
        _emul = caps_sa_amd.CapsLib(os.path.join(EMUL_DIR, "libcaps_sa_emul.so"), "caps_sa_emul_")
    return _emul


_small = None


def emul_small():
    """The same kernel sources compiled with 256-element tiles (64-thread workgroups)."""
    global _small
    if os.environ.get("CAPS_EMUL_REV") == "1":
        return emul_rev(True)
    if _small is None:
        subprocess.check_call(["make", "-s", "-C", EMUL_DIR, "libcaps_sa_emul_small.so"])
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import caps_sa_amd
        _small = caps_sa_amd.CapsLib(os.path.join(EMUL_DIR, "libcaps_sa_emul_small.so"), "caps_sa_emul_")
    return _small


_rev = {}


def emul_rev(small: bool = True):
    """The emulation with the threads of every phase run in DESCENDING order (kernel_lang.h CAPS_EMUL_REVERSE): results must
    not depend on it."""
    name = "libcaps_sa_emul_small_rev.so" if small else "libcaps_sa_emul_rev.so"
    if name not in _rev:
        subprocess.check_call(["make", "-s", "-C", EMUL_DIR, name])
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import caps_sa_amd
        _rev[name] = caps_sa_amd.CapsLib(os.path.join(EMUL_DIR, name), "caps_sa_emul_")
    return _rev[name]
