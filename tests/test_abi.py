"""CPU test: the C-ABI library loads and exports every symbol include/openvm_hip.h declares, and
the Python binding table covers exactly that set. No compute calls (no GPU here)."""
import ctypes
import os
import re


def _header_symbols(pkg):
    with open(os.path.join(pkg.INCLUDE_DIR, "openvm_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ovm_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    assert os.path.exists(pkg.LIB_PATH), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(pkg.LIB_PATH)
    syms = _header_symbols(pkg)
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in openvm_hip.h but not exported"


def test_binding_table_matches_header(pkg):
    assert sorted(pkg.SIGNATURES) == _header_symbols(pkg)
    assert pkg.lib().ovm_hip_abi_version() >= 1


def test_sponge_struct_layout(pkg):
    # ovm_sponge_t: 16 state words + absorb_idx + sample_idx, no padding (sponge.rs:39-48)
    assert ctypes.sizeof(pkg.SpongeState) == 18 * 4
    assert pkg.SpongeState.absorb_idx.offset == 64 and pkg.SpongeState.sample_idx.offset == 68


def test_missing_library_fails_loudly(pkg, monkeypatch):
    import pytest
    monkeypatch.setattr(pkg, "_LIB", None)
    monkeypatch.setattr(pkg, "LIB_PATH", "/nonexistent/libopenvm_hip.so")
    with pytest.raises(ImportError):
        pkg.lib()
