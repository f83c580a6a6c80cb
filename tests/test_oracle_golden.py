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


@pytest.mark.skipif(not CASES, reason="no reference proofs (run tools/gen_golden.rs)")
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
def test_oracle_reproduces_reference_proof(path):
    doc = json.load(open(path))
    orc = oracle_lib.load()
    blob = np.array(doc["pk_blob"], dtype=np.uint32)
    traces = [(t["air_idx"], np.array(t["values"], dtype=np.uint32), t["height"], t["public_values"])
              for t in doc["traces"]]
    want = bytes.fromhex(doc["proof_hex"])
    orc.verify(blob, want)
    got = orc.prove(blob, traces)["proof"]
    assert got == want


def reference_cases():
    """Used by the GPU parity test to run the same fixtures through ovm_hip_prove."""
    return CASES
