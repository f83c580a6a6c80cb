"""GPU parity over the reference's own conformance suite (`backend_test_suite!`, crates/backend-tests/src/lib.rs:1502-1712;
restated in tests/conformance.py): every case proved through the C ABI is byte-identical to the oracle's proof and accepted
by the verifier restatement; every negative case is refused by the library with "non-zero root sum" -- where the reference's
prover returns `NonZeroRootSum` -- and the same thread proves correctly right afterwards. Keys are `StarkEngine::keygen`
keys (every AIR optional), so contexts with ABSENT AIRs are covered as the reference covers them (`optional_air`)."""
import numpy as np
import pytest

import airs
import conformance
import parity_util

pytestmark = pytest.mark.gpu
CASES = conformance.suite()


@pytest.mark.parametrize("name,make", CASES, ids=[n for n, _ in CASES])
def test_conformance_case(dev, pkg, orc, name, make):
    c = make()
    blob = airs.pk_blob(c.params, c.airs, preprocessed=c.preprocessed)
    if c.expect == "ok":
        parity_util.assert_identical_to_oracle(dev, pkg, orc, blob, c.airs, c.traces)
        return
    with pytest.raises(RuntimeError, match="NonZeroRootSum"):
        orc.prove(blob, c.traces)
    pk = pkg.ProvingKey(blob)
    with pytest.raises(pkg.OvmError, match="non-zero root sum"):
        dev.prove(pk, parity_util.upload_traces(dev, pkg, pk, c.airs, c.traces))
    ok = conformance.interactions_single_sender_receiver_happy()
    parity_util.assert_identical_to_oracle(dev, pkg, orc, airs.pk_blob(ok.params, ok.airs), ok.airs, ok.traces)


def test_one_key_three_contexts(dev, pkg, orc):
    """`optional_air` as the reference runs it: ONE device key, three contexts in a row (all AIRs, two of four, a lone
    receiver that must fail), then the first again."""
    cases = [conformance.optional_air(i) for i in (1, 2, 3, 1)]
    blob = airs.pk_blob(cases[0].params, cases[0].airs)
    pk = pkg.ProvingKey(blob)
    for c in cases:
        dtr = parity_util.upload_traces(dev, pkg, pk, c.airs, c.traces)
        if c.expect == "prover_err":
            with pytest.raises(pkg.OvmError, match="non-zero root sum"):
                dev.prove(pk, dtr)
            continue
        got, secs = dev.prove(pk, dtr)
        want = orc.prove(blob, c.traces)["proof"]
        assert got == want, parity_util.first_difference(got, want, secs)
        assert pkg.proof_size(pk, [(t[0], t[2]) for t in c.traces]) == secs


def test_stage_entries_with_absent_airs(dev, pkg, orc):
    """commit / prove_rap_constraints / prove_openings (hal.rs:84-138) on a context with absent AIRs give the sections of
    the monolithic proof; the caller-side coordinator observes a presence flag per optional AIR (prover/mod.rs:155-161)."""
    c = conformance.optional_air(2)
    blob = airs.pk_blob(c.params, c.airs)
    pk = pkg.ProvingKey(blob)
    dtr = parity_util.upload_traces(dev, pkg, pk, c.airs, c.traces)
    mono, secs = dev.prove(pk, dtr)
    pcs = dev.commit_traces(pk, dtr)
    ts = orc.transcript()
    for x in [(7 * i + 1) % airs.P for i in range(8)]:
        ts.observe(x)
    for x in pkg.from_mont(pcs.root):
        ts.observe(int(x))
    by_air = {t[0]: t for t in c.traces}
    for a, air in enumerate(c.airs):
        assert not air["is_required"]
        ts.observe(1 if a in by_air else 0)
        if a in by_air:
            ts.observe(by_air[a][2].bit_length() - 1)
    st, ai, si = ts.state()
    sponge = pkg.SpongeState()
    for i, x in enumerate(pkg.to_mont(st)):
        sponge.state[i] = int(x)
    sponge.absorb_idx, sponge.sample_idx = ai, si
    gkr, batch, r = dev.prove_rap_constraints(pk, dtr, sponge)
    assert gkr == mono[secs[0]:secs[1]] and batch == mono[secs[1]:secs[2]]
    stacking, whir = dev.prove_openings(pk, dtr, pcs, r, sponge)
    assert stacking == mono[secs[2]:secs[3]] and whir == mono[secs[3]:secs[4]]
    pcs.free()
