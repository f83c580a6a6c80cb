"""Multi-AIR proving contexts for the sharded prover's tests: (params, airs, traces, owners) with
traces = (air_idx, common main [canonical, column-major], height, public values, cached mains or None)
and owners[i] = the rank that holds traces[i]. Test infrastructure."""
import numpy as np

import airs


def _bench_partitioned(n_airs, cols, log_heights, params, seed=0):
    air = airs.uniform_runner_air_partitioned(cols)
    traces = []
    for i in range(n_airs):
        h = 1 << log_heights[i]
        traces.append((i, airs.splitmix64_bits(seed + 1000 + i, h), h, [], airs.splitmix64_bits(seed + 2000 + i, cols * h)))
    return params, [air] * n_airs, traces


def _owners_blocks(traces, world):
    """Owners assigned in blocks over the traces in the prover's sorted order (height desc, air index)."""
    order = sorted(range(len(traces)), key=lambda i: (-traces[i][2], traces[i][0]))
    owners = [0] * len(traces)
    for pos, i in enumerate(order):
        owners[i] = pos * world // len(traces)
    return owners


def _fuzz(seed, world):
    """A random multi-AIR context (as tests/fuzz_parity.py draws them) with a random non-decreasing assignment of the
    sorted traces to ranks -- ranks may own several traces or none."""
    rng = np.random.default_rng(seed)
    l_skip = int(rng.choice([0, 1, 2, 3, 4, 1, 2, 3, 4, 5, 6, 9]))
    deg = int(rng.choice([3, 4]))
    params = airs.test_params(l_skip=l_skip, n_stack=max(int(rng.integers(8, 11)) - max(l_skip - 4, 0), 4), k_whir=int(rng.integers(1, 4)))
    params["max_constraint_degree"] = deg
    air_list, traces = [], []
    for i in range(int(rng.integers(1, 7))):
        width = int(rng.integers(2, 20))
        npv = int(rng.integers(0, 3))
        air_list.append(airs.random_air(rng, width, int(rng.integers(0, 60)), int(rng.integers(0, 4)), max_degree=deg,
                                        num_public_values=npv))
        h = 1 << int(rng.integers(0, min(10, l_skip + params["n_stack"]) + 1))
        traces.append((i, rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32), h,
                       [int(x) for x in rng.integers(0, airs.P, size=npv)], None))
    # absent AIRs: optional AIRs lose their traces at random (tests/fuzz_parity.py); with fewer traces than ranks some ranks
    # own nothing -- "a rank whose AIRs are all absent"
    air_list, traces, _ = airs.drop_random_airs(rng, air_list, traces)
    order = sorted(range(len(traces)), key=lambda i: (-traces[i][2], traces[i][0]))
    ranks = np.sort(rng.integers(0, world, size=len(traces)))
    owners = [0] * len(traces)
    for pos, i in enumerate(order):
        owners[i] = int(ranks[pos])
    return params, air_list, traces, owners


def make(name, world):
    if name.startswith("fuzz:"):
        return _fuzz(int(name.split(":")[1]), world)
    if name.startswith("bench_scale:"):  # config D's multi-GPU form at 2^<log_rows> rows x 100 columns per rank, app parameters
        lr = int(name.split(":")[1])
        p, a, t = _bench_partitioned(world, 100, [lr] * world, airs.app_params(lr + 2), seed=lr)
        return p, a, t, _owners_blocks(t, world)
    if name == "bench8":  # one partitioned BenchmarkAir per rank, equal heights (config D's multi-GPU form, small)
        p, a, t = _bench_partitioned(world, 8, [5] * world, airs.test_params(l_skip=2, n_stack=6, k_whir=2))
    elif name == "bench_mixed_heights":  # twice as many AIRs as ranks, heights incl. one below 2^l_skip
        lh = ([6, 5, 5, 4, 3, 3, 1, 0] if world <= 4 else [7, 6, 6, 5, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 0, 0])[:2 * world]
        p, a, t = _bench_partitioned(len(lh), 6, lh, airs.test_params(l_skip=2, n_stack=7, k_whir=2), seed=7)
    elif name == "bench_app":  # app-like parameters (l_skip 4, k_whir 4), 2^10 rows x 20 columns per AIR
        p, a, t = _bench_partitioned(world, 20, [10] * world, airs.test_params(l_skip=4, n_stack=8, k_whir=4), seed=3)
    elif name == "three_airs":  # fewer AIRs than ranks when world = 4: the last rank owns nothing
        p, a, t = _bench_partitioned(3, 8, [5, 4, 4], airs.test_params(l_skip=2, n_stack=6, k_whir=2), seed=11)
    elif name == "mixture":  # Fibonacci (common main only, public values) + partitioned sender/receiver + plain benchmark AIR
        c_airs, c_traces = airs.cached_fixture_11()
        ftr, fpv = airs.fibonacci_trace(0, 1, 16)
        bench = airs.uniform_runner_air(5)
        a = [airs.fibonacci_air()] + c_airs + [bench]
        t = [(0, ftr, 16, fpv, None)] + [(x + 1, c, h, pv, k) for x, c, h, pv, k in c_traces] + \
            [(3, airs.splitmix64_bits(9, 5 * 32), 32, [], None)]
        p = airs.test_params()
    elif name == "no_interactions_on_one_rank":  # the last rank's AIR has no interactions: its leaf run is empty
        p = airs.test_params(l_skip=2, n_stack=6, k_whir=2)
        air = airs.uniform_runner_air_partitioned(8)
        ftr, fpv = airs.fibonacci_trace(0, 1, 4)
        a = [air] * (world - 1) + [airs.fibonacci_air()]
        t = [(i, airs.splitmix64_bits(40 + i, 32), 32, [], airs.splitmix64_bits(50 + i, 8 * 32)) for i in range(world - 1)]
        t.append((world - 1, ftr, 4, fpv, None))
    elif name == "absent_airs":  # a key of six optional AIRs, three of them without a trace (AIR indices 0, 2, 5)
        p = airs.test_params(l_skip=2, n_stack=7, k_whir=2)
        air = dict(airs.uniform_runner_air_partitioned(6), is_required=False)
        lh = {1: 6, 3: 4, 4: 5}
        a = [air] * 6
        t = [(i, airs.splitmix64_bits(70 + i, 1 << lh[i]), 1 << lh[i], [], airs.splitmix64_bits(80 + i, 6 << lh[i])) for i in sorted(lh)]
    elif name.startswith("optional_air:"):  # the reference's `optional_air` contexts 1 and 2 (tests/conformance.py), sharded
        import conformance
        c = conformance.optional_air(int(name.split(":")[1]))
        p, a = c.params, c.airs
        t = [tuple(x) + (None,) * (5 - len(x)) for x in c.traces]
    else:
        raise KeyError(name)
    return p, a, t, _owners_blocks(t, world)


CASES = ["bench8", "bench_mixed_heights", "bench_app", "mixture", "no_interactions_on_one_rank", "absent_airs", "optional_air:1",
         "optional_air:2"]
