"""Memory-bounded modes (reference: GpuProverConfig, cuda-backend/src/device.rs:25-30):
codeword dropped and re-encoded at opening time, round 0 in slabs, capped allocator cache. Proofs must not
change; the allocator's peak must go down."""
import numpy as np
import pytest

import airs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _defaults(pkg):
    yield
    pkg.set_prover_config()  # back to "everything resident"
    pkg._lib().ovm_hip_trim()


def _prove(dev, pkg, pk, traces):
    pkg._lib().ovm_hip_trim()
    pkg.memory_stats(reset_peak=True)
    proof, _ = dev.prove(pk, traces)
    return proof, pkg.memory_stats()["peak"]


def test_modes_keep_the_proof_and_lower_the_peak(dev, pkg, orc):
    """2^16 x 40 BenchmarkAir (app-like parameters, log_stacked_height 18): identical bytes in every mode,
    byte-identical to the oracle, no higher peak with the save-memory modes."""
    params = airs.test_params(l_skip=4, n_stack=14, k_whir=4)
    air = airs.uniform_runner_air(40)
    blob = airs.pk_blob(params, [air])
    pk = pkg.ProvingKey(blob)
    h = 1 << 16
    tr = airs.splitmix64_bits(77, 40 * h)
    d = dev.upload(pkg.to_mont(tr))
    base, peak_base = _prove(dev, pkg, pk, [(0, d, h, [])])
    assert base == orc.prove(blob, [(0, tr, h, [])])["proof"]
    # per-row scratch of round 0: 2 cosets x (3 selectors + 40 columns) x 4 bytes
    assert pkg.constraint_eval_buffer_size(pk, 0) == 2 * 43 * 4
    pkg.set_prover_config(cache_rs_code_matrix=False)
    p1, peak1 = _prove(dev, pkg, pk, [(0, d, h, [])])
    pkg.set_prover_config(zerocheck_save_memory=True, round0_slab_log_rows=10)   # 64 slabs
    p2, peak2 = _prove(dev, pkg, pk, [(0, d, h, [])])
    pkg.set_prover_config(cache_rs_code_matrix=False, zerocheck_save_memory=True, round0_slab_log_rows=12,
                          cache_limit_bytes=64 << 20)
    p3, peak3 = _prove(dev, pkg, pk, [(0, d, h, [])])
    assert p1 == base and p2 == base and p3 == base
    # (at this size the peak is not in round 0; the strict drop is asserted at config E below)
    assert peak1 <= peak_base and peak2 <= peak_base and peak3 <= peak_base
    assert pkg.memory_stats()["cached"] <= 64 << 20
    pkg.set_prover_config(gkr_drop_leaves=True)   # 2^16 rows x 10 interaction slices: a 2^20-leaf tree, base-field numerators
    p4, peak4 = _prove(dev, pkg, pk, [(0, d, h, [])])
    assert p4 == base and peak4 <= peak_base + (1 << 20)   # (the peak is elsewhere at this shape: see the next test)
    print(f"peak bytes: resident {peak_base}, no codeword cache {peak1}, slabs {peak2}, both + capped cache {peak3}, "
          f"leaf layer dropped {peak4}")


def test_dropping_the_leaf_layer_lowers_a_tree_bound_peak(dev, pkg, orc):
    """An AIR whose fraction tree is what peaks (8 columns, 32 interaction pairs, 2^16 rows: a 2^22-leaf tree against a
    2 MB codeword): with gkr_drop_leaves the allocator's peak drops by well over the leaf layer's share, and the proof
    is the oracle's."""
    params = airs.test_params(l_skip=4, n_stack=12, k_whir=4)
    air = airs.uniform_runner_air(8, interactions_per_col=4.0)
    blob = airs.pk_blob(params, [air])
    pk = pkg.ProvingKey(blob)
    h = 1 << 16
    tr = airs.splitmix64_bits(5, 8 * h)
    d = dev.upload(pkg.to_mont(tr))
    base, peak_base = _prove(dev, pkg, pk, [(0, d, h, [])])
    assert base == orc.prove(blob, [(0, tr, h, [])])["proof"]
    pkg.set_prover_config(gkr_drop_leaves=True)
    dropped, peak_drop = _prove(dev, pkg, pk, [(0, d, h, [])])
    assert dropped == base
    print(f"tree-bound AIR: peak {peak_base / 2**20:.1f} MiB resident, {peak_drop / 2**20:.1f} MiB with the leaf layer dropped")
    assert peak_drop < 0.85 * peak_base


def test_leaf_layer_dropped_and_generated_again(dev, pkg, orc):
    """gkr_drop_leaves on the tree shapes the leaf kernels distinguish: a tiny tree with full extension-field numerator
    planes (two dummy-interaction AIRs), several small AIRs whose leaves share one batched launch, mixed heights."""
    import test_gpu_prove as tp
    pkg.set_prover_config(gkr_drop_leaves=True, cache_rs_code_matrix=False)
    tp.test_interactions(dev, pkg, orc)
    tp.test_multi_air_mixed_heights(dev, pkg, orc)
    tp.test_benchmark_air(dev, pkg, orc, 20, 8, 4, 8, 4)


def test_repeated_proofs_make_no_driver_allocation(dev, pkg):
    """Back-to-back proofs of one shape replay the same block splits over the same segments (csrc/block_pool.h): the
    second proof already finds everything in the pool, the pool holds little more than the peak of what was live, and
    ovm_hip_trim gives it all back."""
    params = airs.test_params(l_skip=4, n_stack=14, k_whir=4)
    pk = pkg.ProvingKey(airs.pk_blob(params, [airs.uniform_runner_air(40)]))
    h = 1 << 16
    d = dev.upload(pkg.to_mont(airs.splitmix64_bits(78, 40 * h)))
    pkg._lib().ovm_hip_trim()
    pkg.memory_stats(reset_peak=True)
    first, _ = dev.prove(pk, [(0, d, h, [])])
    after_first = pkg.memory_stats()
    for _ in range(3):
        again, _ = dev.prove(pk, [(0, d, h, [])])
        assert again == first
    m = pkg.memory_stats()
    assert m["driver_allocs"] == after_first["driver_allocs"], "a repeated proof reached hipMalloc"
    assert m["peak_segments"] <= 1.5 * m["peak"] + (64 << 20)   # held from the driver vs handed out at the peak
    held = m["segments"]
    pkg._lib().ovm_hip_trim()
    assert pkg.memory_stats()["segments"] < held and pkg.memory_stats()["cached"] <= 64 << 20   # (partly used small segments stay)


def test_slabs_with_rotation_preprocessed_and_interactions(dev, pkg, orc):
    """Slabs on AIRs whose rows look at their neighbours (rotation wraps across slab ends) and at
    preprocessed / cached partitions."""
    pkg.set_prover_config(cache_rs_code_matrix=False, zerocheck_save_memory=True, round0_slab_log_rows=2)
    params = airs.test_params(l_skip=2, n_stack=6, k_whir=2)
    tr, pvs = airs.fibonacci_trace(0, 1, 64)
    blob = airs.pk_blob(params, [airs.fibonacci_air()])
    got, _ = dev.prove(pkg.ProvingKey(blob), [(0, dev.upload(pkg.to_mont(np.asarray(tr, np.uint32))), 64, pvs)])
    assert got == orc.prove(blob, [(0, tr, 64, pvs)])["proof"]
    sels = [i % 2 == 0 for i in range(32)]
    common, cached = airs.preprocessed_cached_traces(sels, 2)
    pre = {0: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    air = airs.preprocessed_cached_air(2)
    blob = airs.pk_blob(airs.test_params(), [air], preprocessed=pre)
    pk = pkg.ProvingKey(blob)
    up = lambda a: dev.upload(pkg.to_mont(np.asarray(a, np.uint32)))
    handles = [dev.commit_matrix(pk, up(cached[i * 32:(i + 1) * 32]), 1, 32) for i in range(2)]
    got, _ = dev.prove(pk, [(0, up(common), 32, [], handles)])
    assert got == orc.prove(blob, [(0, common, 32, [], cached)])["proof"]


def test_config_e_under_a_capped_cache(dev, pkg, orc):
    """BASELINE config E (2^24 x 100, blowup 4: a 26.8 GB codeword) with the codeword dropped, round 0 in
    slabs of 2^20 rows, the fraction tree's leaf layer dropped and at most 8 GB kept in the allocator's free lists: the verifier accepts the proof
    and the peak is tens of GB below the resident run's."""
    import torch
    params = airs.stress_params(24)
    blob = airs.pk_blob(params, [airs.uniform_runner_air(100)])
    pk = pkg.ProvingKey(blob)
    h = 1 << 24
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(0xD15EA5E)
    trace = torch.randint(0, 2, (100 * h,), dtype=torch.int32, device="cuda:0", generator=gen)
    trace *= int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    resident, peak_res = _prove(dev, pkg, pk, [(0, trace, h, [])])
    pkg.set_prover_config(cache_rs_code_matrix=False, zerocheck_save_memory=True, round0_slab_log_rows=20,
                          cache_limit_bytes=8 << 30, gkr_drop_leaves=True)
    bounded, peak_bnd = _prove(dev, pkg, pk, [(0, trace, h, [])])
    assert bounded == resident
    orc.verify(blob, bounded)
    print(f"config E peak: resident {peak_res / 2**30:.1f} GiB, bounded {peak_bnd / 2**30:.1f} GiB")
    assert peak_bnd < peak_res - (20 << 30)
    del trace
