"""Pins the oracle (and the Python restatement): to Plonky3's published Poseidon2 known-answer vector (always), and to
REAL reference output when it is available.

`tools/gen_golden.rs`, run on a machine with cargo against openvm-org/stark-backend v2.0.1, writes
tests/golden/reference/*.json. While that directory is empty these tests are skipped and the
oracle stays "parity unpinned" (oracle/README.md)."""
import glob
import json
import os

import numpy as np
import pytest

import oracle_lib
import py_ref

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "reference")
CASES = sorted(p for p in glob.glob(os.path.join(REF, "*.json")) if not p.endswith("poseidon2_kat.json"))
KAT = os.path.join(REF, "poseidon2_kat.json")


P3_KAT = os.path.join(HERE, "golden", "poseidon2_p3_kat.json")


def p3_kat():
    import p3_rng
    kat = json.load(open(P3_KAT))
    return kat, p3_rng.poseidon2_constants_from_rng_128(kat["seed"])


def test_p3_known_answer_pins_the_python_restatement():
    """Plonky3's `test_poseidon2_width_16_random`: the explicit-matrix big-int permutation (py_ref), over the constants
    its seeded RNG draws, maps the test's input to the test's expected output. Pins: M_E = circ(2 M4, M4, M4, M4) with
    M4 = [2 3 1 1; ...], M_I = J + diag(V) for THE V the oracle and the product use, x^7, 4 + 13 + 4 rounds with the
    leading external layer, and where each constant is added."""
    kat, (ri, rt, rn) = p3_kat()
    from test_oracle_commit import _p2_consts
    _, _, _, dg = _p2_consts(oracle_lib.load())
    assert py_ref.permute(kat["input"], ri, rt, rn, dg) == kat["output"]


def test_p3_known_answer_pins_the_oracle():
    """The same vector through the ORACLE's own layer code (`poseidon2_permute_rc`; the default permutation is that
    function over the in-tree constant tables, which the reference's poseidon2.cuh pins)."""
    kat, (ri, rt, rn) = p3_kat()
    orc = oracle_lib.load()
    assert list(map(int, orc.permute_rc(np.array(kat["input"], dtype=np.uint32), ri, rt, rn))) == kat["output"]
    # and the default permutation IS permute_rc over the default tables
    from test_oracle_commit import _p2_consts
    di, dt, dn, _ = _p2_consts(orc)
    x = np.arange(16, dtype=np.uint32) * 7919 + 1
    assert np.array_equal(orc.permute_rc(x, di, dt, dn), orc.permute(x))


def test_p3_known_answer_detects_a_wrong_layer():
    """The vector is sensitive to what it claims to pin: a transposed M4, a swapped diagonal entry or one missing round
    all change the output."""
    kat, (ri, rt, rn) = p3_kat()
    from test_oracle_commit import _p2_consts
    _, _, _, dg = _p2_consts(oracle_lib.load())
    bad = list(dg)
    bad[3], bad[4] = bad[4], bad[3]
    assert py_ref.permute(kat["input"], ri, rt, rn, bad) != kat["output"]
    saved = py_ref.M4
    try:
        py_ref.M4 = [list(r) for r in zip(*saved)]
        assert py_ref.permute(kat["input"], ri, rt, rn, dg) != kat["output"]
    finally:
        py_ref.M4 = saved
    assert py_ref.permute(kat["input"], ri, rt, rn[:12] + [0], dg) != kat["output"]


@pytest.mark.skipif(not os.path.exists(KAT), reason="no reference Poseidon2 vector (run tools/gen_golden.rs)")
def test_poseidon2_known_answer():
    kat = json.load(open(KAT))
    orc = oracle_lib.load()
    assert list(map(int, orc.permute(np.array(kat["input"], dtype=np.uint32)))) == kat["output"]
    from test_oracle_commit import _p2_consts
    ri, rt, rn, dg = _p2_consts(orc)
    assert py_ref.permute(np.array(kat["input"], dtype=np.uint32), ri, rt, rn, dg) == kat["output"]


def doc_traces(doc):
    """The traces of a reference document as the oracle / parity helpers take them: (air_idx, common main, height, public
    values, cached partitions concatenated or None)."""
    out = []
    for t in doc["traces"]:
        cached = [v for part in t.get("cached", []) for v in part]
        out.append((t["air_idx"], np.array(t["values"], dtype=np.uint32), t["height"], t["public_values"],
                    np.array(cached, dtype=np.uint32) if cached else None))
    return out


def blob_airs(blob):
    """The per-AIR facts tests/parity_util.upload_traces needs (cached widths), read back from a key blob."""
    w = [int(x) for x in blob]
    at = 12 + w[11] + 8
    n, at = w[at], at + 1
    out = []
    for _ in range(n):
        n_cached = w[at + 6]
        cached = w[at + 7:at + 7 + n_cached]
        at += 7 + n_cached + 1
        n_nodes = w[at]
        at += 1 + 6 * n_nodes
        at += 1 + w[at]
        n_int = w[at]
        at += 1
        for _ in range(n_int):
            at += 4 + w[at + 3]
        out.append({"cached_main_widths": cached})
    return out


@pytest.mark.skipif(not CASES, reason="no reference proofs (run tools/run_on_cargo_box.sh)")
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
def test_oracle_reproduces_reference_proof(path):
    """REAL reference output (tools/gen_golden.rs on a machine with cargo): the oracle must verify the reference's proof
    and reproduce it byte for byte; a negative case must fail in the oracle's prover too."""
    check_document(json.load(open(path)))


def check_document(doc):
    orc = oracle_lib.load()
    blob = np.array(doc["pk_blob"], dtype=np.uint32)
    traces = [t if t[4] is not None else t[:4] for t in doc_traces(doc)]
    if doc.get("proof_hex") is None:
        with pytest.raises(RuntimeError):
            orc.prove(blob, traces)
        return
    want = bytes.fromhex(doc["proof_hex"])
    orc.verify(blob, want)
    got = orc.prove(blob, traces)["proof"]
    assert got == want


@pytest.mark.skipif(not CASES, reason="no reference proofs (run tools/run_on_cargo_box.sh)")
def test_restated_fixtures_equal_the_reference_fixtures():
    """The conformance cases restated in tests/conformance.py against what the reference's own fixtures produced: the same
    key blob apart from the vk pre-hash (which only the reference can compute), the same traces."""
    check_restated_fixtures(CASES)


def check_restated_fixtures(paths):
    import airs
    import conformance
    by_name = {os.path.basename(p)[:-5]: p for p in paths}
    checked = 0
    for name, make in conformance.suite():
        path = by_name.get(name.replace("[", "_").replace("]", "_"))
        if not path:
            continue
        doc, c = json.load(open(path)), make()
        ref = np.array(doc["pk_blob"], dtype=np.uint32)
        nq = int(ref[11])
        pre = [int(x) for x in ref[12 + nq:20 + nq]]
        mine = airs.pk_blob(c.params, c.airs, vk_pre_hash=pre, preprocessed=c.preprocessed)
        assert np.array_equal(mine, ref), f"{name}: the restated key differs from the reference's"
        for (a, tr, h, pv, cached), t in zip(doc_traces(doc), c.traces):
            assert (a, h, list(pv)) == (t[0], t[2], [int(x) for x in t[3]]) and np.array_equal(tr, np.asarray(t[1], dtype=np.uint32)), name
            mine_c = t[4] if len(t) > 4 and t[4] is not None else None
            assert (cached is None) == (mine_c is None) and (cached is None or np.array_equal(cached, np.asarray(mine_c, dtype=np.uint32))), name
        checked += 1
    assert checked, "no reference document matches a conformance case name"


def test_golden_plumbing_on_documents_the_oracle_made(tmp_path):
    """No cargo here, so the consumers above would first run on the day real documents arrive. This writes documents in
    tools/gen_golden.rs's format FROM THE ORACLE (a stand-in, not a pin: same author) for a few conformance cases --
    preprocessed + cached + absent AIRs + a negative case -- under the file names the generator uses, and runs both
    consumers over them."""
    import airs
    import conformance
    orc = oracle_lib.load()
    names = {"mixture_traces_stark[3]", "optional_air[case2]", "optional_air[case3]", "cached_trace_roundtrip[5-5-4]",
             "preprocessed_and_cached_trace_roundtrip[2-8-3-2]"}
    paths = []
    for name, make in conformance.suite():
        if name not in names:
            continue
        c = make()
        pre = [(11 * i + 5) % airs.P for i in range(8)]   # a pre-hash the restatement cannot know
        blob = airs.pk_blob(c.params, c.airs, vk_pre_hash=pre, preprocessed=c.preprocessed)
        traces = []
        for t in c.traces:
            a = c.airs[t[0]]
            cached, off = [], 0
            for w in a["cached_main_widths"]:
                cached.append([int(x) for x in t[4][off:off + w * t[2]]])
                off += w * t[2]
            traces.append({"air_idx": t[0], "height": t[2], "values": [int(x) for x in t[1]],
                           "public_values": [int(x) for x in t[3]], "cached": cached})
        proof_hex, err = None, None
        try:
            proof_hex = orc.prove(blob, c.traces)["proof"].hex()
        except RuntimeError as e:
            err = str(e)
        assert (proof_hex is None) == (c.expect == "prover_err")
        path = tmp_path / (name.replace("[", "_").replace("]", "_") + ".json")
        path.write_text(json.dumps({"fixture": name, "pk_blob": [int(x) for x in blob], "traces": traces,
                                    "proof_hex": proof_hex, "prover_error": err}))
        paths.append(str(path))
        check_document(json.load(open(path)))
    assert len(paths) == len(names)
    check_restated_fixtures(paths)


def reference_cases():
    """Used by the GPU parity test to run the same fixtures through ovm_hip_prove."""
    return CASES
