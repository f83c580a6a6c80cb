"""Synthetic replay of a captured OpenVM workload (the reference's benchmarks/synthetic: 209 proving
segments of a reth block, ~30 AIRs each with the real circuits' widths, constraint counts/degrees,
interaction counts and message lengths). Shapes come from the derived fixture
tests/golden/reth_block_23992138_shapes.json.gz; AIRs are rebuilt as synthetic_air.rs does."""
import numpy as np
import pytest

import airs

SEGS = airs.load_reth_shapes()


def _pick():
    """a small, a Keccak-bearing (widest AIR) and the most populated segment"""
    widest = max(range(len(SEGS)), key=lambda i: max(r[1] for r in SEGS[i]["airs"]))
    most = max(range(len(SEGS)), key=lambda i: len(SEGS[i]["airs"]))
    return sorted({0, widest, most})


def _case(seg_i, max_log_height, params):
    air_list, shapes = airs.synthetic_segment(SEGS[seg_i], max_log_height)
    blob = airs.pk_blob(params, air_list)
    return blob, shapes


def test_fixture_shape():
    assert len(SEGS) == 209 and sum(len(s["airs"]) for s in SEGS) == 6382
    assert all(s["deg"] == 4 for s in SEGS)


def test_oracle_proves_a_clamped_segment(orc):
    params = airs.test_params(l_skip=2, n_stack=8, k_whir=3, max_constraint_degree=4)
    blob, shapes = _case(0, 4, params)
    traces = [(i, np.zeros(w << lh, np.uint32), 1 << lh, []) for i, lh, w in shapes]
    out = orc.prove(blob, traces)
    orc.verify(blob, out["proof"])


@pytest.mark.gpu
@pytest.mark.parametrize("seg_i", _pick())
def test_gpu_matches_oracle_on_clamped_segments(dev, pkg, orc, seg_i):
    """All AIR shapes of a segment at heights <= 2^5: HIP proof byte-identical to the oracle's."""
    import torch
    params = airs.test_params(l_skip=2, n_stack=8, k_whir=3, max_constraint_degree=4)
    blob, shapes = _case(seg_i, 5, params)
    traces = [(i, np.zeros(w << lh, np.uint32), 1 << lh, []) for i, lh, w in shapes]
    want = orc.prove(blob, traces)["proof"]
    orc.verify(blob, want)
    pk = pkg.ProvingKey(blob)
    dtr = [(i, torch.zeros(w << lh, dtype=torch.int32, device="cuda:0"), 1 << lh, []) for i, lh, w in shapes]
    got, _ = dev.prove(pk, dtr)
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("seg_i", _pick()[1:])
def test_gpu_full_size_segments_verify(dev, pkg, orc, seg_i):
    """The same segments at their captured heights (<= 2^22) under app parameters: verifier accepts."""
    import torch
    params = airs.app_params(22)
    params["max_constraint_degree"] = 4  # the captured global degree (the profile predates APP_MAX = 3)
    blob, shapes = _case(seg_i, 22, params)
    pk = pkg.ProvingKey(blob)
    dtr = [(i, torch.zeros(w << lh, dtype=torch.int32, device="cuda:0"), 1 << lh, []) for i, lh, w in shapes]
    proof, _ = dev.prove(pk, dtr)
    orc.verify(blob, proof)


@pytest.mark.gpu
def test_gpu_config_b_shape_verifies(dev, pkg, orc):
    """BASELINE config B shape: one AIR with KeccakfPermAir's captured shape (2 634 columns, 3 183
    degree-3 constraints, two 102-field interactions) at 2^20 rows -- 2.76e9 cells, codeword 22 GB --
    under app parameters. (The real p3-keccak-air constraints are not in the reference tree; this is the
    synthetic stand-in the reference's own synthetic benchmark uses.) The verifier must accept."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs ~150 GB of free HBM")
    rec = next(r for s in SEGS for r in s["airs"] if r[1] == 2634)
    air, width = airs.synthetic_air(rec)
    params = airs.app_params(24)
    blob = airs.pk_blob(params, [air])
    pk = pkg.ProvingKey(blob)
    h = 1 << 20
    trace = torch.zeros(width * h, dtype=torch.int32, device="cuda:0")
    import time
    proof, _ = dev.prove(pk, [(0, trace, h, [])])  # first call: allocations
    orc.verify(blob, proof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    proof2, _ = dev.prove(pk, [(0, trace, h, [])])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"config B shape: {width} x 2^20 = {width * h / 1e9:.2f}e9 cells proved in {ms:.1f} ms "
          f"({width * h / ms / 1e6:.2f}e9 cells/s)")
    assert proof2 == proof
    del trace
    pkg._lib().ovm_hip_trim()
