"""ovm_hip_proof_size (host only): the proof length and its five section ends are a function of the
key and the trace heights; checked here against the lengths of the ORACLE's proofs for the reference's
fixture shapes (the product never proves twice to find a buffer size)."""
import numpy as np
import pytest

import airs


def _check(pkg, orc, params, air_list, traces, preprocessed=None):
    blob = airs.pk_blob(params, air_list, preprocessed=preprocessed)
    out = orc.prove(blob, traces)
    if preprocessed:
        pytest.skip("keys with preprocessed traces commit at load time: needs a device")
    pk = pkg.ProvingKey(blob)
    ends = pkg.proof_size(pk, [(t[0], t[2]) for t in traces])
    assert ends[4] == len(out["proof"]) and ends == sorted(ends)


def test_fibonacci(pkg, orc):
    for log_h in (1, 4, 6):
        tr, pvs = airs.fibonacci_trace(0, 1, 1 << log_h)
        _check(pkg, orc, airs.test_params(l_skip=2, n_stack=6, k_whir=3), [airs.fibonacci_air()], [(0, tr, 1 << log_h, pvs)])


def test_interactions_and_mixed_heights(pkg, orc):
    s = np.array([[0, 1], [7, 4], [3, 5], [546, 889]], dtype=np.uint32)
    r = np.array([[1, 5], [3, 4], [4, 4], [2, 5], [0, 123], [545, 889], [1, 889], [0, 456]], dtype=np.uint32)
    traces = [(0, np.ascontiguousarray(s.T).reshape(-1), 4, []), (1, np.ascontiguousarray(r.T).reshape(-1), 8, [])]
    _check(pkg, orc, airs.test_params(l_skip=2, n_stack=5, k_whir=2),
           [airs.dummy_interaction_air(1, True, 0), airs.dummy_interaction_air(1, False, 0)], traces)
    bench = airs.uniform_runner_air(8)
    t1, pv = airs.fibonacci_trace(0, 1, 16)
    _check(pkg, orc, airs.test_params(l_skip=2, n_stack=6, k_whir=2), [airs.fibonacci_air(), bench, bench],
           [(0, t1, 16, pv), (1, airs.splitmix64_bits(1, 8 * 64), 64, []), (2, airs.splitmix64_bits(2, 8 * 2), 2, [])])


def test_cached_partitions_and_absent_air(pkg, orc):
    c_airs, c_traces = airs.cached_fixture_11()
    _check(pkg, orc, airs.test_params(), c_airs, c_traces)
    # an optional AIR without a trace: its slot in the header is empty
    opt = airs.uniform_runner_air(4)
    opt["is_required"] = False
    tr, pvs = airs.fibonacci_trace(0, 1, 8)
    _check(pkg, orc, airs.test_params(), [airs.fibonacci_air(), opt], [(0, tr, 8, pvs)])


def test_benchmark_air_app_like(pkg, orc):
    params = airs.test_params(l_skip=4, n_stack=8, k_whir=4)
    _check(pkg, orc, params, [airs.uniform_runner_air(20)], [(0, airs.splitmix64_bits(3, 20 << 8), 1 << 8, [])])


def test_rejects_bad_shapes(pkg):
    pk = pkg.ProvingKey(airs.pk_blob(airs.test_params(), [airs.fibonacci_air()]))
    for shapes in ([(0, 3)], [(5, 4)], [(0, 4), (0, 4)]):
        with pytest.raises(pkg.OvmError):
            pkg.proof_size(pk, shapes)
