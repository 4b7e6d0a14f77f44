"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol
include/beom_hip.h declares, and the ctypes struct matches the C struct.  No compute
calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from beom_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "beom_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(beom_\w+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert sorted(capi.EXPORTS) == _declared()


def test_library_exports_every_declared_symbol():
    if not os.path.exists(capi.LIB_PATH):
        pytest.fail("libbeom_hip.so not built: run python -c 'import __graft_entry__ as g; g.build()'")
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.beom_abi_version() == capi.BEOM_ABI_VERSION


def test_library_was_built_from_these_sources():
    """The Makefile embeds a hash of the sources; the binding refuses a stale library."""
    lib = capi.load()
    lib.beom_source_hash.restype = ctypes.c_char_p
    assert lib.beom_source_hash().decode() == capi.source_hash()


def test_params_struct_size_matches_oracle_build():
    import oracle_lib
    oracle_lib.load()          # asserts sizeof(beom_params) == ctypes size


def test_no_cpu_fallback_in_product():
    """The product package must not reference the oracle."""
    pkg = os.path.join(ROOT, "beom_amd")
    for dp, _, fs in os.walk(pkg):
        for fn in fs:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".f95")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "oracle_lib" not in txt and "libbeom_oracle" not in txt, os.path.join(dp, fn)
