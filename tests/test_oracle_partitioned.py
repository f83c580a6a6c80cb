"""Oracle: preprocessed and cached-main traces (the reference's PreprocessedFibFixture,
CachedFixture11, PreprocessedAndCachedFixture; test_utils/mod.rs:261-436)."""
import numpy as np
import pytest

import airs


def _sels(log_h):
    return [i % 2 == 0 for i in range(1 << log_h)]


@pytest.mark.parametrize("log_h", [1, 3, 5])
def test_preprocessed_fib(orc, log_h):
    params = airs.test_params()
    sels = _sels(log_h)
    tr, pvs = airs.fibonacci_selector_trace(0, 1, sels)
    pre = {0: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    blob = airs.pk_blob(params, [airs.fibonacci_selector_air()], preprocessed=pre)
    out = orc.prove(blob, [(0, tr, len(sels), pvs)])
    orc.verify(blob, out["proof"])
    bad = bytearray(out["proof"])
    bad[len(bad) // 2] ^= 1
    with pytest.raises(RuntimeError):
        orc.verify(blob, bytes(bad))


def test_cached_fixture_11(orc):
    params = airs.test_params()
    air_list, traces = airs.cached_fixture_11()
    blob = airs.pk_blob(params, air_list)
    out = orc.prove(blob, traces)
    orc.verify(blob, out["proof"])


@pytest.mark.parametrize("parts", [1, 3])
def test_preprocessed_and_cached(orc, parts):
    params = airs.test_params()
    sels = _sels(4)
    common, cached = airs.preprocessed_cached_traces(sels, parts)
    pre = {0: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    blob = airs.pk_blob(params, [airs.preprocessed_cached_air(parts)], preprocessed=pre)
    out = orc.prove(blob, [(0, common, len(sels), [], cached)])
    orc.verify(blob, out["proof"])


def test_mixture(orc):
    """Fibonacci + partitioned sender/receiver + preprocessed Fibonacci in one proof (MixtureFixture)."""
    params = airs.test_params()
    c_airs, c_traces = airs.cached_fixture_11()
    sels = _sels(3)
    ptr, ppv = airs.fibonacci_selector_trace(0, 1, sels)
    ftr, fpv = airs.fibonacci_trace(0, 1, 16)
    air_list = [airs.fibonacci_air()] + c_airs + [airs.fibonacci_selector_air()]
    pre = {3: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    blob = airs.pk_blob(params, air_list, preprocessed=pre)
    traces = [(0, ftr, 16, fpv)] + [(a + 1, c, h, p, k) for a, c, h, p, k in c_traces] + [(3, ptr, len(sels), ppv)]
    out = orc.prove(blob, traces)
    orc.verify(blob, out["proof"])
