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


# ---- ABI drift check: header vs the Rust binding vs the ctypes table (reference: scripts/gen_stub.py
# generates stubs from the cbindgen header and scripts/abi_cmp.py compares them with the library) ----
import re

_C_TO_RUST = {"void": "c_void", "int": "c_int", "char": "c_char", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8",
              "size_t": "usize", "ovm_stream_t": "OvmStream", "ovm_sponge_t": "OvmSponge", "ovm_pk_t": "OvmPk",
              "ovm_pcs_t": "OvmPcs", "ovm_trace_t": "OvmTrace", "ovm_pcs_params_t": "OvmPcsParams",
              "ovm_matrix_t": "OvmMatrix", "ovm_shard_t": "OvmShard", "ovm_shm_comm_t": "OvmShmComm", "ovm_ipc_comm_t": "OvmIpcComm",
              "ovm_prover_config_t": "OvmProverConfig", "ovm_ipc_status_t": "OvmIpcStatus"}


def _c_type_to_rust(t):
    m = re.match(r"^(const\s+)?(\w+)\s*((?:\*\s*)*)$", t.strip())
    assert m, t
    const, base, stars = m.group(1), m.group(2), m.group(3).replace(" ", "")
    r = _C_TO_RUST[base]
    for i, _ in enumerate(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def header_declarations():
    """{name: (return type, [argument types])} of include/openvm_hip.h, in Rust spelling."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "openvm_hip.h")).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*([A-Za-z_][\w \*]*?)\b(ovm_hip_\w+)\s*\(([^;{]*?)\)\s*;", s, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        types = []
        if args != "void":
            for a in args.split(","):
                am = re.match(r"^(.*?)(\w+)(\[\d+\])?$", a.strip())
                types.append(_c_type_to_rust(am.group(1) + ("*" if am.group(3) else "")))
        out[name] = (None if ret == "void" else _c_type_to_rust(ret), types)
    return out


def rust_declarations():
    """{name: (return type, [argument types])} of crates/hip-backend/src/ffi.rs."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = open(os.path.join(root, "crates", "hip-backend", "src", "ffi.rs")).read()
    block = s[s.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (ovm_hip_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, args, ret = m.group(1), " ".join(m.group(2).split()).rstrip(","), m.group(3)
        types = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()] if args else []
        out[name] = (ret.strip() if ret else None, types)
    return out


def test_rust_binding_matches_header():
    """Every C entry is declared in ffi.rs with the same name, argument count, argument types and return
    type -- and nothing else is."""
    h, r = header_declarations(), rust_declarations()
    assert len(h) >= 45
    assert sorted(h) == sorted(r), set(h) ^ set(r)
    for name in h:
        assert h[name] == r[name], (name, h[name], r[name])


def test_ctypes_table_matches_header(pkg):
    """The Python binding's signature table: same names, same argument counts, and pointer / integer
    classes that agree with the header."""
    import ctypes
    h = header_declarations()
    assert sorted(h) == sorted(pkg.SIGNATURES), set(h) ^ set(pkg.SIGNATURES)

    def klass(rust):
        if rust is None:
            return "void"
        if rust.startswith("*") or rust == "OvmStream":
            return "ptr"
        return {"u32": "u32", "u64": "u64", "usize": "u64", "c_int": "i32", "u8": "u8"}[rust]

    def cklass(ct):
        if ct is None:
            return "void"
        if ct in (ctypes.c_void_p, ctypes.c_char_p) or isinstance(ct, type) and issubclass(ct, ctypes._Pointer):
            return "ptr"
        return {ctypes.c_uint32: "u32", ctypes.c_uint64: "u64", ctypes.c_size_t: "u64", ctypes.c_int: "i32"}[ct]

    for name, (ret, args) in h.items():
        cres, cargs = pkg.SIGNATURES[name]
        assert len(args) == len(cargs), name
        assert klass(ret) == cklass(cres), name
        for a, c in zip(args, cargs):
            assert klass(a) == cklass(c), (name, a, c)


def test_struct_layouts_match_header(pkg):
    """POD structs that cross the boundary: size and field offsets as the C compiler lays them out."""
    import ctypes
    assert ctypes.sizeof(pkg.TraceStruct) == 48 and pkg.TraceStruct.cached.offset == 40
    assert ctypes.sizeof(pkg.PcsParams) == 16 and ctypes.sizeof(pkg.MatrixStruct) == 24
    assert ctypes.sizeof(pkg.ShardStruct) == 48 and pkg.ShardStruct.allgather_device.offset == 24
    assert pkg.ShardStruct.alltoall_device.offset == 32 and pkg.ShardStruct.abort.offset == 40
    assert ctypes.sizeof(pkg.ProverConfigStruct) == 24 and pkg.ProverConfigStruct.cache_limit_bytes.offset == 16
    assert ctypes.sizeof(pkg.IpcStatusStruct) == 232 and pkg.IpcStatusStruct.reason.offset == 40


def test_rust_crate_only_calls_declared_entries():
    """Every `ffi::ovm_hip_*` the (uncompiled) Rust crate calls is declared in its ffi.rs, the key-blob magics match
    the ones the C side parses, and the struct fields the crate touches exist."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "crates", "hip-backend", "src")
    declared = set(rust_declarations())
    used = set()
    for f in os.listdir(src):
        if f.endswith(".rs") and f != "ffi.rs":
            used |= set(re.findall(r"ffi::(ovm_hip_\w+)", open(os.path.join(src, f)).read()))
    assert used and used <= declared, used - declared
    blob_rs = open(os.path.join(src, "pk_blob.rs")).read()
    host_h = open(os.path.join(root, "stark-backend_amd", "csrc", "host.h")).read()
    assert "0x4F56_4D31" in blob_rs and "0x4F564D31u" in host_h
    assert "0x5052_4550" in blob_rs and "0x50524550u" in host_h
    ffi_rs = open(os.path.join(src, "ffi.rs")).read()
    for field in ("absorb_idx", "sample_idx", "d_common_main", "num_cached", "allgather_device", "alltoall_device", "cache_limit_bytes"):
        assert f"pub {field}:" in ffi_rs
