"""The device memory pool (csrc/block_pool.h) stressed over host memory: no GPU needed."""
import pytest


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pool_invariants_hold_under_a_random_workload(pkg, seed):
    pkg.pool_selftest(seed=seed, ops=20000)
