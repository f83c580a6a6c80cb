"""Mechanical audit of the never-compiled Rust crate (crates/hip-backend) against the reference sources it is written
against (the C side has tests/test_abi.py). Without rustc this cannot prove the crate compiles; it
does pin the things a reviewer would otherwise have to check by eye:
  * every `openvm_stark_backend::…` / `openvm_stark_sdk::…` item the crate imports exists, `pub`, in that crate, and the
    module path it is imported through exists;
  * every trait of the reference the crate implements gets exactly the trait's methods -- same names, same number of
    parameters, every method without a default body present -- and its associated types / constants;
  * struct literals of reference types name exactly the struct's fields; associated functions are called with the
    declared number of arguments; methods called on reference types exist.
Reads /root/reference (present where the CPU suite runs; the test is skipped elsewhere)."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CRATE = os.path.join(ROOT, "crates", "hip-backend", "src")
CRATES = {"openvm_stark_backend": "crates/stark-backend/src", "openvm_stark_sdk": "crates/stark-sdk/src"}

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")


def _strip_comments(src):
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _read_all(directory):
    out = {}
    for p in glob.glob(os.path.join(directory, "**", "*.rs"), recursive=True):
        out[p] = _strip_comments(open(p).read())
    return out


@pytest.fixture(scope="module")
def ours():
    return _read_all(CRATE)


@pytest.fixture(scope="module")
def ref():
    return {name: _read_all(os.path.join(REF, d)) for name, d in CRATES.items()}


def _expand_use(tree, prefix=""):
    """`a::{b::{C, D}, e::F}` -> ['a::b::C', 'a::b::D', 'a::e::F']"""
    tree = tree.strip()
    m = re.match(r"^([\w:]*?)(?:::)?\{(.*)\}$", tree, flags=re.S)
    if not m:
        return [prefix + tree] if tree else []
    head, body = m.group(1), m.group(2)
    parts, depth, cur = [], 0, ""
    for ch in body:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    out = []
    for p in parts:
        out += _expand_use(p, prefix + (head + "::" if head else ""))
    return out


def _imports(ours):
    paths = set()
    for src in ours.values():
        for m in re.finditer(r"\buse\s+((?:openvm_stark_backend|openvm_stark_sdk)[^;]*);", src, flags=re.S):
            for p in _expand_use(re.sub(r"\s+", "", m.group(1))):
                paths.add(re.sub(r"\bas\w+$", "", p))
    return sorted(paths)


def _defined_pub(crate_files, name):
    pat = re.compile(r"\bpub(?:\([^)]*\))?\s+(?:unsafe\s+)?(?:struct|enum|trait|type|fn|const|static|mod|use)\b[^;{]*\b%s\b" % re.escape(name))
    macro = re.compile(r"macro_rules!\s+%s\b" % re.escape(name))
    return any(pat.search(s) or macro.search(s) for s in crate_files.values())


def _module_exists(crate_dir, mods):
    base = os.path.join(REF, crate_dir)
    for i in range(len(mods)):
        d = os.path.join(base, *mods[:i + 1])
        if not (os.path.isdir(d) or os.path.exists(d + ".rs")):
            # an inline `pub mod x {` or a re-exported module
            parent = os.path.join(base, *mods[:i])
            cands = [parent + ".rs", os.path.join(parent, "mod.rs"), os.path.join(base, "lib.rs")]
            if not any(os.path.exists(c) and re.search(r"\bpub\s+(?:mod|use)\b[^;]*\b%s\b" % mods[i], _strip_comments(open(c).read())) for c in cands):
                return False
    return True


def test_every_imported_item_exists_in_the_reference(ours, ref):
    paths = _imports(ours)
    assert len(paths) >= 25, paths
    missing = []
    for p in paths:
        parts = p.split("::")
        crate, mods, leaf = parts[0], parts[1:-1], parts[-1]
        if not _defined_pub(ref[crate], leaf):
            missing.append(p + "  (item not defined pub)")
        elif not _module_exists(CRATES[crate], mods):
            missing.append(p + "  (module path)")
    assert not missing, "\n".join(missing)


def _block(src, start):
    """text of the {...} block whose '{' is at or after `start`"""
    i = src.index("{", start)
    depth = 0
    for j in range(i, len(src)):
        if src[j] == "{":
            depth += 1
        elif src[j] == "}":
            depth -= 1
            if depth == 0:
                return src[i + 1:j]
    raise ValueError("unbalanced braces")


def _count_params(sig):
    """number of parameters (incl. self) in the text between the fn's outer parentheses"""
    depth, n, cur = 0, 0, ""
    for ch in sig:
        if ch in "(<[{":
            depth += 1
        elif ch in ")>]}":
            depth -= 1
        if ch == "," and depth == 0:
            n += bool(cur.strip())
            cur = ""
        else:
            cur += ch
    return n + bool(cur.strip())


def _fns(block):
    """{name: (n_params, has_default_body)} of the fns declared at the top level of a trait / impl block"""
    out, depth, i = {}, 0, 0
    while i < len(block):
        ch = block[i]
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        elif depth == 0:
            m = re.compile(r"\bfn\s+(\w+)\s*(?:<[^(]*>)?\s*\(").match(block, i)
            if m:
                j, d = m.end(), 1
                while d:
                    d += {"(": 1, ")": -1}.get(block[j], 0)
                    j += 1
                k = j
                while block[k] not in ";{":
                    k += 1
                out[m.group(1)] = (_count_params(block[m.end():j - 1]), block[k] == "{")
                i = k
                continue
        i += 1
    return out


def _trait(ref_files, name):
    for src in ref_files.values():
        m = re.search(r"\bpub\s+trait\s+%s\b" % name, src)
        if m:
            return _block(src, m.end())
    raise AssertionError(f"trait {name} not found in the reference")


TRAITS = ["ProverBackend", "ProverDevice", "TraceCommitter", "MultiRapProver", "OpeningProver", "DeviceDataTransporter",
          "MatrixDimensions", "FiatShamirTranscript"]


@pytest.mark.parametrize("trait", TRAITS)
def test_trait_impls_match_the_reference_trait(ours, ref, trait):
    tblock = _trait(ref["openvm_stark_backend"], trait)
    tfns = _fns(tblock)
    t_items = set(re.findall(r"^\s*(?:type|const)\s+(\w+)", tblock, flags=re.M))
    found = 0
    for path, src in ours.items():
        for m in re.finditer(r"\bimpl(?:<[^{]*?>)?\s+(?:[\w:]+::)?%s\b[^{;]*\bfor\b[^{]*\{" % trait, src):
            found += 1
            iblock = _block(src, m.start())
            ifns = _fns(iblock)
            for name, (n, _) in ifns.items():
                assert name in tfns, f"{os.path.basename(path)}: `{name}` is not a method of {trait}"
                assert n == tfns[name][0], f"{os.path.basename(path)}: {trait}::{name} takes {tfns[name][0]} parameters, the impl has {n}"
            for name, (_, has_default) in tfns.items():
                assert has_default or name in ifns, f"{os.path.basename(path)}: {trait}::{name} (no default body) is not implemented"
            i_items = set(re.findall(r"^\s*(?:type|const)\s+(\w+)", iblock, flags=re.M))
            assert i_items <= t_items, f"{trait}: unknown associated items {i_items - t_items}"
            required = {x for x in t_items if not re.search(r"(?:type|const)\s+%s\b[^;]*=" % x, tblock)}
            assert required <= i_items, f"{trait}: associated items not given: {required - i_items}"
    assert found, f"the crate does not implement {trait}"


def _struct_fields(ref_files, name):
    for src in ref_files.values():
        m = re.search(r"\bpub\s+struct\s+%s\b[^;{(]*\{" % name, src)
        if m:
            return set(re.findall(r"^\s*(?:pub(?:\([^)]*\))?\s+)?(\w+)\s*:", _block(src, m.start()), flags=re.M))
    raise AssertionError(f"struct {name} not found in the reference")


@pytest.mark.parametrize("name", ["CommittedTraceData", "DeviceStarkProvingKey"])
def test_struct_literals_name_the_reference_fields(ours, ref, name):
    fields = _struct_fields(ref["openvm_stark_backend"], name)
    seen = 0
    for path, src in ours.items():
        for m in re.finditer(r"\b%s\s*\{" % name, src):
            if re.search(r"(struct|impl|for)\s*$", src[max(0, m.start() - 12):m.start()]):
                continue
            body = _block(src, m.start())
            used = set(re.findall(r"(?:^|,)\s*(\w+)\s*(?=:|,|$)", re.sub(r"\{.*?\}|\(.*?\)", "", body, flags=re.S).strip()))
            used.discard("")
            assert used == fields, f"{os.path.basename(path)}: {name} {{ {sorted(used)} }} vs reference fields {sorted(fields)}"
            seen += 1
    assert seen, f"no literal of {name} in the crate"


CALLS = [  # (type, associated fn, number of arguments at the crate's call sites)
    ("DeviceMultiStarkProvingKey", "new", 5), ("Coordinator", "new", 3), ("ColMajorMatrix", "new", 2),
]


@pytest.mark.parametrize("ty,fn,nargs", CALLS)
def test_associated_functions_are_called_with_the_declared_arity(ours, ref, ty, fn, nargs):
    decl = None
    for src in ref["openvm_stark_backend"].values():
        for m in re.finditer(r"\bimpl(?:<[^{]*?>)?\s+%s\b[^{]*\{" % ty, src):
            fns = _fns(_block(src, m.start()))
            if fn in fns:
                decl = fns[fn][0]
        if decl is None and fn == "new":  # `#[derive(derive_new::new)]`: one parameter per field, in field order
            m = re.search(r"#\[derive\([^\]]*derive_new::new[^\]]*\)\]\s*pub\s+struct\s+%s\b[^;{(]*\{" % ty, src)
            if m:
                decl = len(re.findall(r"^\s*(?:pub(?:\([^)]*\))?\s+)?\w+\s*:", _block(src, m.start()), flags=re.M))
    assert decl == nargs, f"{ty}::{fn} declares {decl} parameters"
    calls = 0
    for path, src in ours.items():
        for m in re.finditer(r"\b%s(?:::<[^(]*>)?::%s\s*\(" % (ty, fn), src):
            j, d = m.end(), 1
            while d:
                d += {"(": 1, ")": -1}.get(src[j], 0)
                j += 1
            assert _count_params(src[m.end():j - 1]) == nargs, f"{os.path.basename(path)}: {ty}::{fn} call"
            calls += 1
    assert calls, f"{ty}::{fn} is never called"


METHODS = {"openvm_stark_backend": ["mat_view", "to_matrix", "decode_from_bytes", "encode_to_vec", "default_from_params", "k_whir"],
           "openvm_stark_sdk": ["poseidon2_perm", "keygen_and_prove"]}


def test_methods_called_on_reference_types_exist(ours, ref):
    everything = "\n".join(ours.values()) + open(os.path.join(ROOT, "crates", "hip-backend", "tests", "integration.rs")).read() \
        + open(os.path.join(ROOT, "tools", "gen_golden.rs")).read()
    for crate, names in METHODS.items():
        for n in names:
            if re.search(r"\b%s\s*(?:::<[^(]*>)?\(" % n, everything):
                hay = dict(ref["openvm_stark_backend"])
                hay.update(ref["openvm_stark_sdk"])
                assert any(re.search(r"\bfn\s+%s\b" % n, s) for s in hay.values()), f"`fn {n}` not found in the reference"
