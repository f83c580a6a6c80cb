"""Host-side check of the cached-column program form (csrc/program.h: linear_terms, stage_cached -- what the round-0 and
GKR-leaf kernels of the HIP path execute, csrc/cached_vm.cuh). `ovm_hip_staged_selftest` compiles the zerocheck, LogUp and
GKR-leaf programs of every AIR of a key, takes the accumulation roots apart into linear terms at several support caps, stages
every form with a given image (rows shared by cached columns and live values; columns brought per block) and runs it on a
random row next to the plain register-program interpreter and a direct evaluation of the DAG, entirely on the host: block
cutting, the row pool, eviction, operand normalisation (constants on the left, commuted forwards, RSUB / MOVC), accumulation
words, LEAF markers at group ends are all exercised without a GPU. (The GPU parity tests then pin the kernels.)"""
import numpy as np
import pytest

import airs


def _random_key(seed, n_airs=4):
    rng = np.random.default_rng(seed)
    air_list = []
    for _ in range(n_airs):
        width = int(rng.integers(2, 40))
        air_list.append(airs.random_air(rng, width, int(rng.integers(1, 60)), int(rng.integers(0, 5)),
                                        num_public_values=int(rng.integers(0, 3))))
    return airs.pk_blob(airs.test_params(), air_list)


@pytest.mark.parametrize("stage_rows,acc_rows", [(2, 1), (2, 3), (3, 2), (4, 4), (8, 8), (16, 16), (64, 64)])
def test_staged_form_equals_plain_program_on_random_airs(pkg, stage_rows, acc_rows):
    total = 0
    for seed in range(6):
        total += pkg.staged_selftest(_random_key(100 + seed), stage_rows, acc_rows, seed=seed + 1)
    assert total > 0
    if stage_rows <= 4:
        assert total > 200  # tiny layouts: programs span many blocks


def test_staged_form_on_fixture_airs(pkg):
    """The reference-derived fixture AIRs (Fibonacci, interactions, selectors/preprocessed, cached,
    BenchmarkAir) and the Keccak-shaped synthetic AIR of the captured reth workload."""
    air_list = [airs.fibonacci_air(), airs.dummy_interaction_air(2, True, 0), airs.dummy_interaction_air(3, False, 0),
                airs.uniform_runner_air(20), airs.fibonacci_selector_air()]
    pre = {4: (np.array([i % 2 == 0 for i in range(16)], np.uint32), 16)}
    blob = airs.pk_blob(airs.test_params(), air_list, preprocessed=pre)
    for layout in [(2, 1), (4, 4), (8, 8)]:
        assert pkg.staged_selftest(blob, *layout) > 0
    segs = airs.load_reth_shapes()
    rec = next(r for s in segs for r in s["airs"] if r[1] == 2634)
    air, _ = airs.synthetic_air(rec)
    params = airs.app_params(22)
    params["max_constraint_degree"] = 4
    blocks = pkg.staged_selftest(airs.pk_blob(params, [air]), 8, 8)
    assert blocks > 400  # 3 183 accumulations, at most 8 per block


def test_host_transcript_permutation_matches_scalar(pkg):
    """The Fiat-Shamir transcript's host permutation is dispatched at run time (AVX2 where available,
    csrc/host_poseidon2.cpp); it must equal the scalar Poseidon2 on every state."""
    assert pkg._lib().ovm_hip_host_permute_selftest(20000, 7) == 1


def test_host_gkr_batch_arithmetic_matches_scalar(pkg):
    """The host side of the LogUp-GKR layer tails (pair sums, folds, eq weights over batches of records) is vectorised
    with AVX2 where the CPU has it (csrc/host_gkr_batch.cpp); it must equal the scalar code on every batch size."""
    assert pkg._lib().ovm_hip_host_gkr_batch_selftest(20, 11) == 1
