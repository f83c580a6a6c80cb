"""Pins the oracle (and the Python restatement) to REAL reference output when it is available.

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
