"""The reference's conformance suite (tests/conformance.py = crates/backend-tests/src/lib.rs sections 2, 3, 5, 6) through
the ORACLE on the CPU: every positive case proves and its proof is accepted by the verifier restatement; every negative
case fails in the prover with `NonZeroRootSum` exactly where the reference's does
(prover/logup_zerocheck/fractional_sumcheck_gkr.rs:88-91). No GPU; the `-m gpu` twin compares the HIP prover with these
proofs byte for byte."""
import numpy as np
import pytest

import airs
import conformance

CASES = conformance.suite()


@pytest.mark.parametrize("name,make", CASES, ids=[n for n, _ in CASES])
def test_oracle_conformance(orc, name, make):
    c = make()
    blob = airs.pk_blob(c.params, c.airs, preprocessed=c.preprocessed)
    if c.expect == "prover_err":
        with pytest.raises(RuntimeError, match="NonZeroRootSum"):
            orc.prove(blob, c.traces)
        return
    out = orc.prove(blob, c.traces)
    orc.verify(blob, out["proof"])
    bad = bytearray(out["proof"])
    bad[len(bad) // 2] ^= 1
    with pytest.raises(RuntimeError):
        orc.verify(blob, bytes(bad))


def test_every_air_of_an_engine_key_is_optional_and_params_follow_the_reference():
    c = conformance.optional_air(2)
    assert all(not a["is_required"] for a in c.airs) and [t[0] for t in c.traces] == [1, 3]
    p = conformance.default_test_params_small()
    # test_system_params_small(2, 8, 3): log_final_poly_len 10 % 3 = 1 -> 3 WHIR rounds; 4 protocol bits: unique
    # decoding at rate 1/2 gives ceil(4 / -log2(0.75)) = 10 queries, list decoding (m = 3) at rates 2^-3, 2^-5
    # ceil(4 / -log2(sqrt(rho) * 7 / 6)) = 4 and 2
    assert p["whir_num_queries"] == [10, 4, 2] and (p["mu_pow_bits"], p["folding_pow_bits"], p["query_phase_pow_bits"]) == (3, 2, 1)
    assert conformance.fib_roundtrip_params(2)["whir_num_queries"] == [191, 87]  # 79 bits, unique decoding: ceil(79 / -log2((1 + rho) / 2)) at rho = 2^-1, 2^-4


def test_self_interaction_air_shape():
    """SelfInteractionAir: 8 interactions over `width`-field messages; the 'next' counts reuse the LOCAL sum
    (self_interaction_air.rs:39-40), so the DAG holds one sum chain and its negation."""
    a = conformance.self_interaction_air(4, 4)
    assert len(a["interactions"]) == 8 and a["constraint_idx"] == [] and a["need_rot"]
    assert all(len(msg) == 4 and cw == 1 and bus == 4 for bus, cw, _, msg in a["interactions"])
    cnt = [c for _, _, c, _ in a["interactions"]]
    assert cnt[2] == cnt[6] and cnt[3] == cnt[7] and a["nodes"][cnt[3]] == (airs.N_NEG, cnt[2])
    tr, h = conformance.self_interaction_trace(7, 3)
    assert h == 8 and tr.reshape(7, 8)[2, 6] == (6 + 2) % 7
