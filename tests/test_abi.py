"""CPU: the product library builds for gfx950, loads, and exports every symbol that
include/caps_sa_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "caps_sa_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(caps_sa_hip_\w+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("caps_sa_hip_build_u32", "caps_sa_hip_build_u64", "caps_sa_hip_build_device_u32",
                 "caps_sa_hip_verify_device_u32", "caps_sa_hip_merge_u32", "caps_sa_hip_sort_suffixes_u32",
                 "caps_sa_hip_upper_bound_u32", "caps_sa_hip_lcp_u32", "caps_sa_hip_last_error"):
        assert must in names


def test_library_builds_and_exports_every_declared_symbol():
    import caps_sa_amd
    path = caps_sa_amd.build_library()
    assert os.path.exists(path)
    lib = caps_sa_amd.lib()
    for name in _declared():
        assert hasattr(lib.dll, name), f"{name} declared in include/caps_sa_hip.h but not exported"
    assert "gfx950" in lib.version()
    assert sorted("caps_sa_hip_" + e for e in caps_sa_amd.EXPORTS) == _declared()


def test_argument_errors_without_gpu():
    import numpy as np
    import caps_sa_amd
    lib = caps_sa_amd.lib()
    T = np.frombuffer(b"ACGT" * 10, dtype=np.uint8)
    with pytest.raises(caps_sa_amd.CapsSaError) as e:
        lib.build(T[:20], max_context=5)               # bounded context below the reference's domain (n < 32): checked before any GPU work
    assert e.value.code == -2
    assert lib.workspace_bytes(1 << 20, 64, 32) > (1 << 20) * 32


def test_product_package_has_no_oracle_or_emulation_dependency():
    pkg = os.path.join(ROOT, "caps-sa_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(pkg, "libcaps_sa_hip.so")]).decode()
    assert "caps_sa_emul_" not in out and "caps_oracle_" not in out
