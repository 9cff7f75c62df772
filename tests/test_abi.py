"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/agents_amd.h declares (no kernel launches here)."""
import ctypes
import os
import re

from agents_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "agents_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aa_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_covers_header(lib):
    assert declared_symbols() == _lib.exported_symbols()


def test_abi_version(lib):
    assert lib.aa_abi_version() == 5


def test_argument_validation_without_gpu(lib):
    """Entry points reject bad arguments before touching the device."""
    assert lib.aa_rb_sample_rows(None, 1, 1, 1, 1, 0, 0, None, None, None, None, None) == -22
    assert lib.aa_counter_add(None, 1, None) == -22
    d = _lib.GemmDesc()
    assert lib.aa_gemm_f32_workspace_bytes(ctypes.byref(d)) == -1
    assert lib.aa_colsum_workspace_bytes(0, 4) == -1
    assert lib.aa_colsum_workspace_bytes(1000, 32) > 0


def test_no_product_import_of_oracle():
    """The oracle is test infrastructure: nothing under agents_amd/ may import it."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "agents_amd")):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
