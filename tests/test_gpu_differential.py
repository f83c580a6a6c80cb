"""Differential testing on randomised AIRs and random (constraint-violating) traces: the HIP proof
must equal the oracle's byte for byte. Neither prover checks validity, so every intermediate value
-- constraint evaluations on cosets, LogUp numerators/denominators, sumcheck polynomials -- takes
generic field values instead of the zeros a satisfied AIR produces."""
import numpy as np
import pytest

import airs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_oracle_debug_asserts(monkeypatch):
    # the oracle mirrors a reference debug_assert on the zerocheck sum; traces here violate constraints
    monkeypatch.setenv("ORC_SKIP_DEBUG_ASSERTS", "1")


def _run(dev, pkg, orc, params, air_list, traces):
    blob = airs.pk_blob(params, air_list)
    want = orc.prove(blob, traces)["proof"]
    pk = pkg.ProvingKey(blob)
    dtr = [(a, dev.upload(pkg.to_mont(np.asarray(tr, dtype=np.uint32))), h, pv) for a, tr, h, pv in traces]
    got, secs = dev.prove(pk, dtr)
    if got != want:
        n = min(len(got), len(want))
        first = next((i for i in range(n) if got[i] != want[i]), n)
        pytest.fail(f"proof differs at byte {first} of {len(want)} (section ends {secs})")


@pytest.mark.parametrize("seed", range(20))
def test_random_single_air(dev, pkg, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    l_skip = int(rng.integers(1, 5))
    log_h = int(rng.integers(0, 8))
    n_stack = max(int(rng.integers(3, 8)), log_h + 1 - l_skip if log_h + 1 > l_skip else 3)
    params = airs.test_params(l_skip=l_skip, n_stack=n_stack, k_whir=int(rng.integers(1, 4)))
    width = int(rng.integers(2, 10))
    npv = int(rng.integers(0, 3))
    air = airs.random_air(rng, width, int(rng.integers(1, 12)), int(rng.integers(0, 4)), num_public_values=npv)
    h = 1 << log_h
    tr = rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32)
    pvs = [int(x) for x in rng.integers(0, airs.P, size=npv)]
    _run(dev, pkg, orc, params, [air], [(0, tr, h, pvs)])


@pytest.mark.parametrize("seed", range(10))
def test_random_multi_air(dev, pkg, orc, seed, monkeypatch):
    rng = np.random.default_rng(2000 + seed)
    params = airs.test_params(l_skip=int(rng.integers(2, 4)), n_stack=7, k_whir=2)
    air_list, traces = [], []
    for i in range(int(rng.integers(2, 5))):
        width = int(rng.integers(2, 8))
        air_list.append(airs.random_air(rng, width, int(rng.integers(0, 8)), int(rng.integers(0, 3))))
        h = 1 << int(rng.integers(0, 7))
        traces.append((i, rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32), h, []))
    _run(dev, pkg, orc, params, air_list, traces)


def test_internal_like_parameters(dev, pkg, orc):
    """l_skip 2, log_blowup 3 (the reference's internal-aggregation shape, config/mod.rs:170-199), k_whir 4."""
    params = airs.make_params(2, 9, 3, 4, 3, 5, 2, 2, 2, 2, 3, 512)
    rng = np.random.default_rng(7)
    air = airs.random_air(rng, 6, 8, 2)
    h = 1 << 7
    tr = rng.integers(0, airs.P, size=6 * h, dtype=np.uint64).astype(np.uint32)
    _run(dev, pkg, orc, params, [air], [(0, tr, h, [])])


@pytest.mark.parametrize("stage,acc,rows", [(2, 1, 1), (2, 1, 4), (3, 2, 2), (4, 8, 1), (8, 3, 2), (5, 5, 4)])
@pytest.mark.parametrize("seed", range(4))
def test_round0_interpreter_layouts(dev_hooks, pkg_hooks, orc, monkeypatch, stage, acc, rows, seed):
    """The cached-column round-0 interpreter (program.h stage_cached, csrc/cached_vm.cuh) under forced layouts (test twin of
    the library): the LDS image at the program's minimum or a few rows above it, 2..8 columns brought per block (a block
    boundary every instruction or two), 1 / 2 / 4 trace rows per lane. Wider AIRs with more constraints than the cases
    above so that programs span many blocks."""
    dev, pkg = dev_hooks, pkg_hooks
    monkeypatch.setenv("OVM_STAGE_ROWS", str(stage))
    monkeypatch.setenv("OVM_ACC_ROWS", str(acc))
    monkeypatch.setenv("OVM_R0_ROWS", str(rows))
    rng = np.random.default_rng(3000 + seed)
    l_skip = int(rng.integers(2, 5))
    log_h = int(rng.integers(l_skip, 11))
    params = airs.test_params(l_skip=l_skip, n_stack=max(8, log_h + 1 - l_skip), k_whir=2)
    width = int(rng.integers(6, 24))
    air = airs.random_air(rng, width, int(rng.integers(10, 40)), int(rng.integers(1, 4)), num_public_values=2)
    h = 1 << log_h
    tr = rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32)
    pvs = [int(x) for x in rng.integers(0, airs.P, size=2)]
    _run(dev, pkg, orc, params, [air], [(0, tr, h, pvs)])


@pytest.mark.parametrize("live", [20, 40, 70, 200])
def test_many_live_slots(dev, pkg, orc, monkeypatch, live):
    """Programs with many simultaneously live intermediates: `live` products are created by the first
    constraint and consumed by the second, so the register program needs that many slots. 20 fits
    the staged VM's default LDS image, 40 its last-resort 56 KiB image, 70 and 200 fall back to the plain
    interpreter (slots in private memory, NS 128 / 512).
    (Round 5: round 0's image is one pool of rows for cached columns and live values; on this short trace -- one row per
    lane -- it may take up to 128 KiB, so 20, 40 and 70 run on the cached interpreter and 200 on the plain one.)"""
    rng = np.random.default_rng(4000 + live)
    width = 8
    a = airs.Air("ManyLiveSlots", width)
    prods = [a.local(i % width) * a.local((i + 3) % width) + airs.Expr.lift(i + 1) for i in range(live)]
    s1 = prods[0]
    for p in prods[1:]:
        s1 = s1 + p
    a.assert_zero(s1)
    s2 = prods[0] * a.next(0)
    for i, p in enumerate(prods[1:], 1):
        s2 = s2 + p * a.next(i % width)
    a.assert_zero(s2)
    a.push_interaction(0, [prods[1], a.local(2)], a.local(0))
    a.push_interaction(0, [prods[1], a.local(2)], airs.NEG_ONE() * a.local(0))
    air = a.finish()
    params = airs.test_params(l_skip=3, n_stack=8, k_whir=2)
    h = 1 << 9
    tr = rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32)
    _run(dev, pkg, orc, params, [air], [(0, tr, h, [])])


@pytest.mark.parametrize("lds", ["1", "0"])
@pytest.mark.parametrize("seed", range(4))
def test_unbatched_mle_round_kernels(dev_hooks, pkg_hooks, orc, monkeypatch, lds, seed):
    """The kernels of the LARGE multilinear rounds on small traces: with OVM_MLE_BATCH=0 every (trace, program) job is its
    own launch, through the LDS-slot kernel (default) or the private-memory kernel programs with many slots fall back to
    (OVM_MLE_LDS=0). Degree-3 and degree-4 keys."""
    dev, pkg = dev_hooks, pkg_hooks
    monkeypatch.setenv("OVM_MLE_BATCH", "0")
    monkeypatch.setenv("OVM_MLE_LDS", lds)
    rng = np.random.default_rng(5000 + seed)
    l_skip = int(rng.integers(2, 4))
    params = airs.test_params(l_skip=l_skip, n_stack=9, k_whir=2)
    if seed % 2:
        params["max_constraint_degree"] = 4
    air_list, traces = [], []
    for i in range(3):
        width = int(rng.integers(3, 12))
        air_list.append(airs.random_air(rng, width, int(rng.integers(4, 30)), int(rng.integers(0, 3)),
                                        max_degree=params["max_constraint_degree"]))
        h = 1 << int(rng.integers(l_skip + 1, 11))
        traces.append((i, rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32), h, []))
    _run(dev, pkg, orc, params, air_list, traces)


@pytest.mark.parametrize("degree", [1, 2])
@pytest.mark.parametrize("seed", range(2))
def test_mle_executor_low_degree_keys(dev_hooks, pkg_hooks, orc, monkeypatch, degree, seed):
    """Keys of constraint degree 1 and 2: the multilinear rounds' assembly executor has one statement per number of points
    (csrc/mle_vm_asm.inc, D = 1..4) and the other GPU cases only reach D = 3 and 4. Unbatched, so that the traces of 2^4..2^10
    rows all go through it. (Degree 1: no interactions -- a LogUp constraint has degree 2.)"""
    dev, pkg = dev_hooks, pkg_hooks
    monkeypatch.setenv("OVM_MLE_BATCH", "0")
    rng = np.random.default_rng(7700 + 10 * degree + seed)
    l_skip = 2
    params = airs.test_params(l_skip=l_skip, n_stack=9, k_whir=2, max_constraint_degree=degree)
    air_list, traces = [], []
    for i in range(3):
        width = int(rng.integers(3, 10))
        air_list.append(airs.random_air(rng, width, int(rng.integers(4, 24)), 0 if degree == 1 else int(rng.integers(0, 3)),
                                        max_degree=degree))
        h = 1 << int(rng.integers(l_skip + 2, 11))
        traces.append((i, rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32), h, []))
    _run(dev, pkg, orc, params, air_list, traces)


@pytest.mark.parametrize("seed", range(6))
def test_round0_program_groups(dev_hooks, pkg_hooks, orc, monkeypatch, seed):
    """Round 0 with the program split into root groups that run as extra workgroups of the batched job
    (the path long programs take on short traces), forced here for every program with at least two
    groups; also with two rows per lane and a minimal image."""
    dev, pkg = dev_hooks, pkg_hooks
    monkeypatch.setenv("OVM_R0_GROUPS_MIN", "2")
    if seed % 2:
        monkeypatch.setenv("OVM_R0_ROWS", "2")
        monkeypatch.setenv("OVM_STAGE_ROWS", "3")
        monkeypatch.setenv("OVM_ACC_ROWS", "2")
    rng = np.random.default_rng(6000 + seed)
    l_skip = int(rng.integers(2, 5))
    params = airs.test_params(l_skip=l_skip, n_stack=9, k_whir=2)
    air_list, traces = [], []
    for i in range(3):
        width = int(rng.integers(4, 16))
        air_list.append(airs.random_air(rng, width, int(rng.integers(8, 50)), int(rng.integers(1, 4)), num_public_values=1))
        h = 1 << int(rng.integers(l_skip + 1, 11))
        traces.append((i, rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32), h,
                       [int(rng.integers(0, airs.P))]))
    _run(dev, pkg, orc, params, air_list, traces)


@pytest.mark.parametrize("lazy", ["1", "0"])
def test_whir_w_accumulate_lazy_and_reduced(dev_hooks, pkg_hooks, orc, monkeypatch, lazy):
    """WHIR's  w += sum_q gamma_q eq(., z_q)  with the products accumulated unreduced in 96-bit words (stages.hip:
    whir_w_accumulate_lazy_kernel, taken when the folded polynomial has >= 2^12 coefficients) and with every product
    reduced (OVM_LAZY_W=0): the same proof as the oracle's either way."""
    dev, pkg = dev_hooks, pkg_hooks
    monkeypatch.setenv("OVM_LAZY_W", lazy)
    rng = np.random.default_rng(9200)
    params = airs.test_params(l_skip=3, n_stack=12, k_whir=2)
    air_list, traces = [], []
    for i in range(2):
        width = int(rng.integers(3, 8))
        air_list.append(airs.random_air(rng, width, int(rng.integers(4, 12)), int(rng.integers(1, 3))))
        h = 1 << int(rng.integers(11, 14))
        traces.append((i, rng.integers(0, airs.P, size=width * h, dtype=np.uint64).astype(np.uint32), h, []))
    _run(dev, pkg, orc, params, air_list, traces)
