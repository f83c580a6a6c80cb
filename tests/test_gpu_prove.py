"""GPU parity: the whole prove path through the C ABI produces proof bytes identical to the CPU
oracle's (which the oracle's restatement of the reference verifier accepts)."""
import numpy as np
import pytest

import airs

pytestmark = pytest.mark.gpu
SECTIONS = ["header", "gkr", "batch_constraints", "stacking", "whir"]


def run_case(dev, pkg, orc, params, air_list, traces):
    blob = airs.pk_blob(params, air_list)
    want = orc.prove(blob, traces)
    orc.verify(blob, want["proof"])
    pk = pkg.ProvingKey(blob)
    dtr = [(a, dev.upload(pkg.to_mont(np.asarray(tr, dtype=np.uint32))), h, pv) for a, tr, h, pv in traces]
    got, secs = dev.prove(pk, dtr)
    if got != want["proof"]:
        n = min(len(got), len(want["proof"]))
        first = next((i for i in range(n) if got[i] != want["proof"][i]), n)
        sec = next((SECTIONS[k] for k, e in enumerate(secs) if first < e), "tail")
        pytest.fail(f"proof differs at byte {first} (section {sec}); lengths {len(got)} vs {len(want['proof'])}; sections {secs}")
    orc.verify(blob, got)


@pytest.mark.parametrize("log_h,l_skip,n_stack,k_whir", [(4, 2, 4, 2), (6, 2, 6, 3), (5, 3, 5, 2), (1, 2, 3, 1)])
def test_fibonacci(dev, pkg, orc, log_h, l_skip, n_stack, k_whir):
    params = airs.test_params(l_skip=l_skip, n_stack=n_stack, k_whir=k_whir)
    tr, pvs = airs.fibonacci_trace(0, 1, 1 << log_h)
    run_case(dev, pkg, orc, params, [airs.fibonacci_air()], [(0, tr, 1 << log_h, pvs)])


def test_interactions(dev, pkg, orc):
    params = airs.test_params(l_skip=2, n_stack=5, k_whir=2)
    s = np.array([[0, 1], [7, 4], [3, 5], [546, 889]], dtype=np.uint32)
    r = np.array([[1, 5], [3, 4], [4, 4], [2, 5], [0, 123], [545, 889], [1, 889], [0, 456]], dtype=np.uint32)
    traces = [(0, np.ascontiguousarray(s.T).reshape(-1), 4, []), (1, np.ascontiguousarray(r.T).reshape(-1), 8, [])]
    run_case(dev, pkg, orc, params, [airs.dummy_interaction_air(1, True, 0), airs.dummy_interaction_air(1, False, 0)], traces)


@pytest.mark.parametrize("cols,log_h,l_skip,n_stack,k_whir", [(8, 5, 2, 6, 2), (20, 8, 4, 8, 4), (100, 10, 4, 12, 4)])
def test_benchmark_air(dev, pkg, orc, cols, log_h, l_skip, n_stack, k_whir):
    params = airs.test_params(l_skip=l_skip, n_stack=n_stack, k_whir=k_whir)
    h = 1 << log_h
    tr = airs.splitmix64_bits(0xD15EA5E, cols * h)
    run_case(dev, pkg, orc, params, [airs.uniform_runner_air(cols)], [(0, tr, h, [])])


def test_multi_air_mixed_heights(dev, pkg, orc):
    params = airs.test_params(l_skip=2, n_stack=6, k_whir=2)
    bench = airs.uniform_runner_air(8)
    t1, pv = airs.fibonacci_trace(0, 1, 16)
    t2 = airs.splitmix64_bits(1, 8 * 64)
    t3 = airs.splitmix64_bits(2, 8 * 2)
    run_case(dev, pkg, orc, params, [airs.fibonacci_air(), bench, bench],
             [(0, t1, 16, pv), (1, t2, 64, []), (2, t3, 2, [])])


# ---- full BASELINE sizes: size-independent properties (the oracle cannot prove these in seconds,
# but its verifier restatement checks any proof in milliseconds) -----------------------------------
def _config_d(pkg, dev, log_rows, cols, log_stacked, zero=False):
    import torch
    h = 1 << log_rows
    params = airs.app_params(log_stacked)
    blob = airs.pk_blob(params, [airs.uniform_runner_air(cols)])
    pk = pkg.ProvingKey(blob)
    if zero:
        trace = torch.zeros(cols * h, dtype=torch.int32, device="cuda:0")
    else:
        gen = torch.Generator(device="cuda:0")
        gen.manual_seed(0xD15EA5E)
        bits = torch.randint(0, 2, (cols * h,), dtype=torch.int32, device="cuda:0", generator=gen)
        trace = bits * int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    return blob, pk, trace, h


@pytest.mark.parametrize("zero", [False, True], ids=["random-bits", "all-zero"])
def test_config_d_full_size_verifies(dev, pkg, orc, zero):
    """BASELINE config D (2^22 x 100, app params, log_stacked_height 24): the proof is accepted by the
    verifier, proving twice gives identical bytes, and a flipped byte in any section is rejected."""
    blob, pk, trace, h = _config_d(pkg, dev, 22, 100, 24, zero)
    proof, secs = dev.prove(pk, [(0, trace, h, [])])
    orc.verify(blob, proof)
    proof2, _ = dev.prove(pk, [(0, trace, h, [])])
    assert proof2 == proof, "prove is not deterministic"
    starts = [0] + list(secs[:-1])
    for name, a, b in zip(SECTIONS, starts, secs):
        if b - a < 8:
            continue
        bad = bytearray(proof)
        bad[(a + b) // 2] ^= 1
        with pytest.raises(Exception):
            orc.verify(blob, bytes(bad))


def test_config_c_shape_verifies(dev, pkg, orc):
    """BASELINE config C shape: 4 AIRs x 2^20 rows x 100 columns with LogUp in one proof."""
    import torch
    params = airs.app_params(24)
    air = airs.uniform_runner_air(100)
    blob = airs.pk_blob(params, [air, air, air, air])
    pk = pkg.ProvingKey(blob)
    h = 1 << 20
    one = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    traces = []
    for i in range(4):
        gen = torch.Generator(device="cuda:0")
        gen.manual_seed(0xD15EA5E + i)
        traces.append((i, torch.randint(0, 2, (100 * h,), dtype=torch.int32, device="cuda:0", generator=gen) * one, h, []))
    proof, _ = dev.prove(pk, traces)
    orc.verify(blob, proof)


def test_gkr_three_point_fallback(dev_hooks, pkg_hooks, orc, monkeypatch):
    """The GKR sub-rounds normally take two evaluation points from the device and derive the third
    from the running claim; the fallback for degenerate challenges must give the same bytes (forced through the test
    twin of the library)."""
    dev, pkg = dev_hooks, pkg_hooks
    monkeypatch.setenv("OVM_GKR_THREE_POINTS", "1")
    params = airs.test_params(l_skip=4, n_stack=8, k_whir=4)
    h = 1 << 8
    tr = airs.splitmix64_bits(0xD15EA5E, 20 * h)
    run_case(dev, pkg, orc, params, [airs.uniform_runner_air(20)], [(0, tr, h, [])])


@pytest.mark.parametrize("three_points", [False, True])
def test_gkr_layer_sumchecks_on_split_tables_and_mixed_heights(dev_hooks, pkg_hooks, orc, monkeypatch, three_points):
    """The GKR layer sumchecks on a tall single AIR (base-field leaf numerators, split eq tables from 2^3 records on: forced
    through the test twin), on mixed heights with a padding run that starts on an odd pair, and with the three-point
    fallback."""
    dev, pkg = dev_hooks, pkg_hooks
    if three_points:
        monkeypatch.setenv("OVM_GKR_THREE_POINTS", "1")
    monkeypatch.setenv("OVM_GKR_SPLIT_MIN", "3")
    params = airs.test_params(l_skip=4, n_stack=10, k_whir=4)
    h = 1 << 12
    run_case(dev, pkg, orc, params, [airs.uniform_runner_air(24)], [(0, airs.splitmix64_bits(0xA11CE, 24 * h), h, [])])
    monkeypatch.delenv("OVM_GKR_SPLIT_MIN")
    bench, wide = airs.uniform_runner_air(8), airs.uniform_runner_air(20)
    t1, pv = airs.fibonacci_trace(0, 1, 16)
    run_case(dev, pkg, orc, airs.test_params(l_skip=2, n_stack=9, k_whir=2), [airs.fibonacci_air(), bench, wide, bench],
             [(0, t1, 16, pv), (1, airs.splitmix64_bits(1, 8 * 512), 512, []), (2, airs.splitmix64_bits(2, 20 * 256), 256, []),
              (3, airs.splitmix64_bits(3, 8 * 2), 2, [])])


def test_gkr_tree_built_three_levels_per_pass(dev, pkg, orc):
    """The fraction tree is built three levels per pass (stages.hip: gkr_build3_kernel -- parents and grandparents written,
    never read back; the one or two layers left under the root one level per launch). The oracle's bytes on: a tall
    single AIR (base-field leaf numerators: the PBASE form; tree heights 0, 1 and 2 mod 3), mixed heights whose leaf runs
    start and end on odd indices (parents / grandparents without a sibling in the range: the edge lanes), a tree of a
    single interaction on a two-row trace."""
    for n_stack, cols, h in ((10, 24, 1 << 12), (9, 12, 1 << 11), (8, 12, 1 << 10)):
        params = airs.test_params(l_skip=4, n_stack=n_stack, k_whir=4)
        run_case(dev, pkg, orc, params, [airs.uniform_runner_air(cols)], [(0, airs.splitmix64_bits(0xB01D + cols, cols * h), h, [])])
    bench, wide = airs.uniform_runner_air(8), airs.uniform_runner_air(20)
    t1, pv = airs.fibonacci_trace(0, 1, 16)
    run_case(dev, pkg, orc, airs.test_params(l_skip=2, n_stack=9, k_whir=2), [airs.fibonacci_air(), bench, wide, bench],
             [(0, t1, 16, pv), (1, airs.splitmix64_bits(1, 8 * 512), 512, []), (2, airs.splitmix64_bits(2, 20 * 256), 256, []),
              (3, airs.splitmix64_bits(3, 8 * 2), 2, [])])
    run_case(dev, pkg, orc, airs.test_params(l_skip=1, n_stack=4, k_whir=1), [bench], [(0, airs.splitmix64_bits(5, 8 * 2), 2, [])])


def sponge_before_openings(pkg, orc, raw):
    import ctypes
    s = pkg.SpongeState()
    ctypes.memmove(ctypes.byref(s), raw, len(raw))
    return s


def test_stage_level_entries_match_monolithic(dev, pkg, orc):
    """commit / prove_rap_constraints / prove_openings (the three hal.rs traits) driven by a
    coordinator on the caller's side -- here prover/mod.rs:110-175 restated on the oracle's
    transcript -- give exactly the sections of the monolithic ovm_hip_prove proof."""
    params = airs.test_params(l_skip=2, n_stack=6, k_whir=2)
    bench = airs.uniform_runner_air(8)
    air_list = [airs.fibonacci_air(), bench, bench]
    t1, pv = airs.fibonacci_trace(0, 1, 16)
    traces = [(0, t1, 16, pv), (1, airs.splitmix64_bits(1, 8 * 64), 64, []), (2, airs.splitmix64_bits(2, 8 * 2), 2, [])]
    blob = airs.pk_blob(params, air_list)
    pk = pkg.ProvingKey(blob)
    dtr = [(a, dev.upload(pkg.to_mont(np.asarray(tr, dtype=np.uint32))), h, p) for a, tr, h, p in traces]
    final = pkg.SpongeState()
    mono, secs = dev.prove(pk, dtr, sponge=None)
    orc.verify(blob, mono)

    pcs = dev.commit_traces(pk, dtr)
    root_words = pcs.root.copy()
    root = pkg.from_mont(pcs.root)
    assert root.astype("<u4").tobytes() == mono[4:36]
    ts = orc.transcript()
    for x in [(7 * i + 1) % airs.P for i in range(8)]:  # vk_pre_hash of airs.pk_blob
        ts.observe(x)
    for x in root:
        ts.observe(int(x))
    by_air = {a: (h, p) for a, _, h, p in traces}
    for a, air in enumerate(air_list):
        if not air["is_required"]:
            ts.observe(1 if a in by_air else 0)
        if a in by_air:
            h, p = by_air[a]
            ts.observe(h.bit_length() - 1)
            for x in p:
                ts.observe(int(x))
    st, ai, si = ts.state()
    sponge = pkg.SpongeState()
    for i, x in enumerate(pkg.to_mont(st)):
        sponge.state[i] = int(x)
    sponge.absorb_idx, sponge.sample_idx = ai, si

    gkr, batch, r = dev.prove_rap_constraints(pk, dtr, sponge)
    ts_state = bytes(sponge)
    assert gkr == mono[secs[0]:secs[1]]
    assert batch == mono[secs[1]:secs[2]]
    stacking, whir = dev.prove_openings(pk, dtr, pcs, r, sponge)
    assert stacking == mono[secs[2]:secs[3]]
    assert whir == mono[secs[3]:secs[4]]
    pcs.free()
    # TraceCommitter::commit(&[&Matrix]) as the trait has it: matrices (height-sorted), parameters, no key
    by_h = sorted(range(3), key=lambda i: (-traces[i][2], traces[i][0]))
    mats = [(dtr[i][1], air_list[traces[i][0]]["common_main_width"], traces[i][2]) for i in by_h]
    pcs2 = dev.commit_matrices(params, mats)
    assert (pcs2.root == root_words).all()
    stacking2, whir2 = dev.prove_openings(pk, dtr, pcs2, r, sponge_before_openings(pkg, orc, ts_state))
    assert stacking2 == stacking and whir2 == whir
    with pytest.raises(pkg.OvmError, match="sorted"):
        dev.commit_matrices(params, mats[::-1])
    pcs2.free()
    del final


import test_oracle_golden as _golden


@pytest.mark.skipif(not _golden.reference_cases(), reason="no reference proofs (run tools/run_on_cargo_box.sh)")
@pytest.mark.parametrize("path", _golden.reference_cases())
def test_reference_golden_proofs(dev, pkg, path):
    """When real reference output is present: ovm_hip_prove must reproduce it byte for byte (a negative case must be
    refused)."""
    import json
    import parity_util
    doc = json.load(open(path))
    blob = np.array(doc["pk_blob"], dtype=np.uint32)
    pk = pkg.ProvingKey(blob)
    dtr = parity_util.upload_traces(dev, pkg, pk, _golden.blob_airs(blob), _golden.doc_traces(doc))
    if doc.get("proof_hex") is None:
        with pytest.raises(pkg.OvmError):
            dev.prove(pk, dtr)
        return
    got, _ = dev.prove(pk, dtr)
    assert got == bytes.fromhex(doc["proof_hex"])


# ---- preprocessed / cached-main traces (reference fixtures test_utils/mod.rs:261-436) ----
def run_partitioned(dev, pkg, orc, params, air_list, traces, preprocessed=None):
    """traces: (air_idx, common, height, pvs[, cached partitions concatenated])."""
    blob = airs.pk_blob(params, air_list, preprocessed=preprocessed)
    want = orc.prove(blob, traces)
    orc.verify(blob, want["proof"])
    pk = pkg.ProvingKey(blob)
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
    got, secs = dev.prove(pk, dtr)
    if got != want["proof"]:
        n = min(len(got), len(want["proof"]))
        first = next((i for i in range(n) if got[i] != want["proof"][i]), n)
        sec = next((SECTIONS[k] for k, e in enumerate(secs) if first < e), "tail")
        pytest.fail(f"proof differs at byte {first} (section {sec}); lengths {len(got)} vs {len(want['proof'])}")
    orc.verify(blob, got)


@pytest.mark.parametrize("log_h", [1, 3, 5])
def test_preprocessed_fib(dev, pkg, orc, log_h):
    sels = [i % 2 == 0 for i in range(1 << log_h)]
    tr, pvs = airs.fibonacci_selector_trace(0, 1, sels)
    pre = {0: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    run_partitioned(dev, pkg, orc, airs.test_params(), [airs.fibonacci_selector_air()], [(0, tr, len(sels), pvs)], pre)


def test_cached_fixture_11(dev, pkg, orc):
    air_list, traces = airs.cached_fixture_11()
    run_partitioned(dev, pkg, orc, airs.test_params(), air_list, traces)


@pytest.mark.parametrize("parts", [1, 3])
def test_preprocessed_and_cached(dev, pkg, orc, parts):
    sels = [i % 2 == 0 for i in range(16)]
    common, cached = airs.preprocessed_cached_traces(sels, parts)
    pre = {0: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    run_partitioned(dev, pkg, orc, airs.test_params(), [airs.preprocessed_cached_air(parts)],
                    [(0, common, len(sels), [], cached)], pre)


def test_mixture_fixture(dev, pkg, orc):
    c_airs, c_traces = airs.cached_fixture_11()
    sels = [i % 2 == 0 for i in range(8)]
    ptr, ppv = airs.fibonacci_selector_trace(0, 1, sels)
    ftr, fpv = airs.fibonacci_trace(0, 1, 16)
    air_list = [airs.fibonacci_air()] + c_airs + [airs.fibonacci_selector_air()]
    pre = {3: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    traces = [(0, ftr, 16, fpv)] + [(a + 1, c, h, p, k) for a, c, h, p, k in c_traces] + [(3, ptr, len(sels), ppv)]
    run_partitioned(dev, pkg, orc, airs.test_params(), air_list, traces, pre)


def test_interactions_with_l_skip_zero(dev, pkg, orc):
    """`test_interactions_roundtrip_with_l_skip_zero` (crates/cuda-backend/src/tests.rs:562-573): InteractionsFixture11
    under test_system_params_small(0, 8, 3) -- no univariate skip domain at all: byte-identical to the oracle."""
    params = airs.test_params(l_skip=0, n_stack=8, k_whir=3)
    air_list = [airs.dummy_interaction_air(1, True, 0), airs.dummy_interaction_air(1, False, 0)]
    traces = [(0, np.array([0, 3, 7, 546, 1, 5, 4, 889], np.uint32), 4, []),
              (1, np.array([1, 3, 4, 2, 0, 545, 1, 0, 5, 4, 4, 5, 123, 889, 889, 456], np.uint32), 8, [])]
    run_case(dev, pkg, orc, params, air_list, traces)
    tr, pv = airs.fibonacci_trace(0, 1, 64)
    run_case(dev, pkg, orc, params, [airs.fibonacci_air(), airs.uniform_runner_air(8)],
             [(0, tr, 64, pv), (1, airs.splitmix64_bits(3, 8 * 32), 32, [])])


@pytest.mark.parametrize("l_skip", [0, 6, 9])
def test_mixture_fixture_other_skip_domains(dev, pkg, orc, l_skip):
    """`test_mixture_fixture_gpu_roundtrip_large_l_skip` (crates/cuda-backend/src/tests.rs:609-624: l_skip 6 and 9,
    n_stack = 13 - l_skip) and the degenerate l_skip = 0: cached, preprocessed, interaction and plain traces, every
    one of them SHORTER than the skip domain for l_skip 6 and 9 (lifted traces)."""
    c_airs, c_traces = airs.cached_fixture_11()
    sels = [i % 2 == 0 for i in range(8)]
    ptr, ppv = airs.fibonacci_selector_trace(0, 1, sels)
    ftr, fpv = airs.fibonacci_trace(0, 1, 16)
    air_list = [airs.fibonacci_air()] + c_airs + [airs.fibonacci_selector_air()]
    pre = {3: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    traces = [(0, ftr, 16, fpv)] + [(a + 1, c, h, p, k) for a, c, h, p, k in c_traces] + [(3, ptr, len(sels), ppv)]
    run_partitioned(dev, pkg, orc, airs.test_params(l_skip=l_skip, n_stack=13 - l_skip if l_skip else 8, k_whir=3), air_list, traces, pre)


def test_blowup4_degree4_parity(dev, pkg, orc):
    """Leaf-like parameters (log_blowup 2, global constraint degree 4 -> 3 zerocheck / 4 logup cosets)
    at a size the oracle proves in seconds: byte-exact."""
    params = airs.stress_params(12)
    params["logup_pow_bits"] = 4
    params["query_phase_pow_bits"] = 4
    params["mu_pow_bits"] = 4
    h = 1 << 10
    tr = airs.splitmix64_bits(0xD15EA5E, 20 * h)
    run_case(dev, pkg, orc, params, [airs.uniform_runner_air(20)], [(0, tr, h, [])])


def test_config_e_full_size_verifies(dev, pkg, orc):
    """BASELINE config E: 2^24 rows x 100 columns, blowup 4 (codeword 26.8 GB, 2^29-leaf LogUp tree):
    the proof must be accepted by the verifier restatement."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs ~200 GB of free HBM")
    params = airs.stress_params(24)
    blob = airs.pk_blob(params, [airs.uniform_runner_air(100)])
    pk = pkg.ProvingKey(blob)
    h = 1 << 24
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(0xD15EA5E)
    trace = torch.randint(0, 2, (100 * h,), dtype=torch.int32, device="cuda:0", generator=gen)
    trace *= int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    proof, _ = dev.prove(pk, [(0, trace, h, [])])
    orc.verify(blob, proof)
    del trace
    pkg._lib().ovm_hip_trim()


# ---- error behaviour through the C ABI: invalid inputs give error codes and messages, never crashes ----
def test_error_paths(dev, pkg, orc):
    import torch
    params = airs.test_params(l_skip=2, n_stack=5, k_whir=2)
    air_list = [airs.dummy_interaction_air(1, True, 0), airs.dummy_interaction_air(1, False, 0)]
    pk = pkg.ProvingKey(airs.pk_blob(params, air_list))
    up = lambda a: dev.upload(pkg.to_mont(np.asarray(a, dtype=np.uint32)))
    s = np.array([[0, 1], [7, 4], [3, 5], [546, 889]], dtype=np.uint32).T.reshape(-1)
    # 1. interactions that do not balance (a sender without its receiver): LogUp root sum is non-zero
    with pytest.raises(pkg.OvmError, match="non-zero root sum"):
        dev.prove(pk, [(0, up(s), 4, [])])
    # 2. AIR index out of range
    with pytest.raises(pkg.OvmError, match="air index"):
        dev.prove(pk, [(7, up(s), 4, [])])
    # 3. height not a power of two
    with pytest.raises(pkg.OvmError):
        dev.prove(pk, [(0, up(np.zeros(6, np.uint32)), 3, [])])
    # 4. wrong number of public values
    pkf = pkg.ProvingKey(airs.pk_blob(params, [airs.fibonacci_air()]))
    tr, pvs = airs.fibonacci_trace(0, 1, 16)
    with pytest.raises(pkg.OvmError, match="public values"):
        dev.prove(pkf, [(0, up(tr), 16, pvs[:2])])
    # 5. a cached partition is declared by the AIR but not supplied
    c_airs, c_traces = airs.cached_fixture_11()
    pkc = pkg.ProvingKey(airs.pk_blob(airs.test_params(), c_airs))
    with pytest.raises(pkg.OvmError, match="cached"):
        dev.prove(pkc, [(a, up(c), h, p) for a, c, h, p, _ in c_traces])
    # 6. malformed key blobs
    blob = airs.pk_blob(params, air_list)
    with pytest.raises(pkg.OvmError):
        pkg.ProvingKey(blob[:-3])
    bad = blob.copy()
    bad[0] ^= 1
    with pytest.raises(pkg.OvmError):
        pkg.ProvingKey(bad)
    # 7. proof buffer too small: refused before any work, with the required length (= ovm_hip_proof_size)
    r = np.array([[1, 5], [3, 4], [4, 4], [2, 5], [0, 123], [545, 889], [1, 889], [0, 456]], dtype=np.uint32).T.reshape(-1)
    good = [(0, up(s), 4, []), (1, up(r), 8, [])]
    import ctypes
    arr, keep = pkg._trace_structs(good)
    out = np.zeros(64, dtype=np.uint8)
    plen = ctypes.c_uint64(0)
    sponge = pkg.SpongeState()
    before = bytes(sponge)
    rc = pkg._lib().ovm_hip_prove(pk.handle, len(good), arr, ctypes.byref(sponge), out.ctypes.data, 64,
                                  ctypes.byref(plen), None, dev.stream)
    assert rc == 1 and plen.value > 64 and bytes(sponge) == before  # size reported, transcript untouched
    assert b"too small" in pkg._lib().ovm_hip_last_error()
    assert pkg.proof_size(pk, [(0, 4), (1, 8)])[4] == plen.value
    proof, secs = dev.prove(pk, good)  # the library is still usable after every failure above
    assert len(proof) == plen.value and secs == pkg.proof_size(pk, [(0, 4), (1, 8)])
    # 8. null trace pointer, duplicate AIR index, a word >= P among the public values: error codes, no faults
    with pytest.raises(pkg.OvmError, match="null"):
        dev.prove(pk, [(0, None, 4, []), (1, up(r), 8, [])])
    with pytest.raises(pkg.OvmError, match="duplicate"):
        dev.prove(pk, [(0, up(s), 4, []), (0, up(s), 4, [])])
    arrf, keepf = pkg._trace_structs([(0, up(tr), 16, pvs)])
    keepf[0][0] = 0x78000001  # = P: not a reduced Montgomery word
    outf = np.zeros(1 << 20, dtype=np.uint8)
    rc = pkg._lib().ovm_hip_prove(pkf.handle, 1, arrf, None, outf.ctypes.data, outf.size, ctypes.byref(plen), None, dev.stream)
    assert rc == 1 and b">= P" in pkg._lib().ovm_hip_last_error()
    orc.verify(airs.pk_blob(params, air_list), proof)
    # 9. the degenerate context the sharded fuzzer found in round 4: l_skip = 0, every trace ONE row tall, no interactions ->
    # xi is empty; the reference indexes xi[0] and panics (prover/logup_zerocheck/mod.rs:255): oracle and library both refuse
    p0 = airs.test_params(l_skip=0, n_stack=8, k_whir=1)
    blob0 = airs.pk_blob(p0, [airs.fibonacci_air()])
    tr1, pv1 = airs.fibonacci_trace(0, 1, 1)
    with pytest.raises(RuntimeError, match="EmptyXi"):
        orc.prove(blob0, [(0, tr1, 1, pv1)])
    with pytest.raises(pkg.OvmError, match="xi is empty"):
        dev.prove(pkg.ProvingKey(blob0), [(0, up(tr1), 1, pv1)])
    tr2, pv2 = airs.fibonacci_trace(0, 1, 2)   # two rows: one variable, fine
    run_case(dev, pkg, orc, p0, [airs.fibonacci_air()], [(0, tr2, 2, pv2)])


def test_two_host_threads_prove_concurrently(dev, pkg, orc):
    """One stream per OS thread, one shared proving key (the reference's supported concurrency:
    cuda-backend/examples/keccakf.rs:88-131): both threads must reproduce the single-threaded bytes."""
    import threading
    import torch
    params = airs.test_params(l_skip=4, n_stack=9, k_whir=4)
    air = airs.uniform_runner_air(20)
    blob = airs.pk_blob(params, [air])
    pk = pkg.ProvingKey(blob)
    h = 1 << 11
    tr = dev.upload(pkg.to_mont(airs.splitmix64_bits(5, 20 * h)))
    ref, _ = dev.prove(pk, [(0, tr, h, [])])
    orc.verify(blob, ref)
    torch.cuda.synchronize()
    errors = []

    def worker():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                d = pkg.Device(0)
                for _ in range(6):
                    got, _ = d.prove(pk, [(0, tr, h, [])])
                    if got != ref:
                        errors.append("proof differs")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_thread_shutdown_releases_and_the_thread_keeps_working(dev, pkg, orc):
    """ovm_hip_thread_shutdown: the calling thread's mailboxes, pinned arena, grind scratch and free pool segments are
    released now; the next proof creates them again and is the same proof."""
    params = airs.test_params(l_skip=2, n_stack=6, k_whir=2)
    blob = airs.pk_blob(params, [airs.uniform_runner_air(8)])
    pk = pkg.ProvingKey(blob)
    tr = dev.upload(pkg.to_mont(airs.splitmix64_bits(5, 8 * 32)))
    a, _ = dev.prove(pk, [(0, tr, 32, [])])
    before = pkg.memory_stats()
    assert pkg._lib().ovm_hip_thread_shutdown() == 0
    after = pkg.memory_stats()
    assert after["cached"] == 0 and after["cached"] <= before["cached"]
    b, _ = dev.prove(pk, [(0, tr, 32, [])])
    assert a == b
    orc.verify(blob, b)


def test_failure_while_the_early_lde_is_in_flight(dev, pkg, orc):
    """Round 0's coset LDE is forked onto the library's side stream when the GKR tree is enqueued; the LogUp root check
    comes right after. A proof that fails there (interactions that do not balance) must wait for the side stream before
    its buffers go back to the pool: the next proofs on the same thread are byte-identical to the oracle's."""
    params = airs.test_params(l_skip=3, n_stack=14, k_whir=2)
    h = 1 << 16
    send = airs.dummy_interaction_air(3, True, 0)
    pk_bad = pkg.ProvingKey(airs.pk_blob(params, [send]))
    tr = np.arange(4 * h, dtype=np.uint32) % 9973 + 1  # counts >= 1: a sender without a receiver
    d = dev.upload(pkg.to_mont(tr))
    for _ in range(3):
        with pytest.raises(pkg.OvmError, match="non-zero root sum"):
            dev.prove(pk_bad, [(0, d, h, [])])
        cols = 12
        good = airs.splitmix64_bits(0xFEED, cols * h)
        run_case(dev, pkg, orc, params, [airs.uniform_runner_air(cols)], [(0, good, h, [])])
