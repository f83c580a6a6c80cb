"""Shared by the `-m gpu` parity tests: prove one context with the HIP prover through the C ABI and compare it with the
oracle's proof byte for byte (test infrastructure)."""
import numpy as np
import pytest

SECTIONS = ["header", "gkr", "batch_constraints", "stacking", "whir"]


def upload_traces(dev, pkg, pk, air_list, traces):
    """traces: (air_idx, common main, height, public values[, cached partitions concatenated]) with canonical values;
    cached partitions are committed one by one with ovm_hip_commit_matrix (CommittedTraceData, prover/types.rs:18-25)."""
    dtr = []
    for t in traces:
        a, tr, h, pv = t[:4]
        cached = []
        if len(t) > 4 and t[4] is not None:
            off = 0
            for w in air_list[a]["cached_main_widths"]:
                part = dev.upload(pkg.to_mont(np.asarray(t[4][off:off + w * h], dtype=np.uint32)))
                cached.append(dev.commit_matrix(pk, part, w, h))
                off += w * h
        dtr.append((a, dev.upload(pkg.to_mont(np.asarray(tr, dtype=np.uint32))), h, pv, cached))
    return dtr


def first_difference(got, want, secs):
    n = min(len(got), len(want))
    first = next((i for i in range(n) if got[i] != want[i]), n)
    sec = next((SECTIONS[k] for k, e in enumerate(secs) if first < e), "tail")
    return f"proof differs at byte {first} (section {sec}); lengths {len(got)} vs {len(want)}; sections {secs}"


def assert_identical_to_oracle(dev, pkg, orc, blob, air_list, traces):
    want = orc.prove(blob, traces)["proof"]
    orc.verify(blob, want)
    pk = pkg.ProvingKey(blob)
    got, secs = dev.prove(pk, upload_traces(dev, pkg, pk, air_list, traces))
    if got != want:
        pytest.fail(first_difference(got, want, secs))
    orc.verify(blob, got)
    return got
