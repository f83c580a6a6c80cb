"""The reference's engine-generic conformance suite, restated case by case (test infrastructure).

`backend_test_suite!(Engine)` (crates/backend-tests/src/lib.rs:1502-1712) is the acceptance gate every backend of the
reference runs; `crates/hip-backend/tests/integration.rs` wires it to `HipEngine` for a machine with cargo. This module
restates its prove-path cases -- sections 2 (end to end), 3 (round trips under custom parameters), 5 (custom contexts)
and 6 (interactions, positive and negative) -- as data: parameters, AIRs, traces, and what the reference expects of the
prover. The GPU tests prove each case through the C ABI and compare with the oracle byte for byte
(tests/test_gpu_conformance.py); the CPU tests run the oracle and its verifier restatement over the same cases
(tests/test_oracle_conformance.py).

Two things follow the reference to the letter here where tests/airs.py's older helpers only follow its spirit:
  * keys come from `StarkEngine::keygen` (engine.rs:95-110), which adds every AIR with `add_air` -> `is_required =
    false` (keygen/mod.rs:58-60): every AIR observes a presence flag in the transcript (prover/mod.rs:155-161) and may
    be absent from a context (`optional_air`);
  * parameters are `test_system_params_small` (test_utils/mod.rs:531-579: log_final_poly_len = (n_stack + l_skip) %
    k_whir, pow bits 1 / 2 / 3, LogUp pow 2, 5 bits of security, `SplitUniqueList {m: 3, list_start_round: 1}` -- the
    number of queries of WHIR rounds >= 1 comes from the list-decoding bound, config.rs:262-277,364-372).

A case is `Case(params, airs, traces, preprocessed, expect)`:
  traces  [(air_idx, column-major canonical values, height, public values[, cached partitions concatenated])]
  expect  "ok"           the reference proves and verifies
          "prover_err"   `engine.prove` must fail: LogUp's root sum is non-zero
                         (prover/logup_zerocheck/fractional_sumcheck_gkr.rs:88-91 `NonZeroRootSum`)
"""
import collections
import math

import numpy as np

import airs
from airs import Air, Expr, NEG_ONE, ONE, P

Case = collections.namedtuple("Case", "params airs traces preprocessed expect")


# ---- parameters ---------------------------------------------------------------------------------------------------
def whir_num_queries(log_blowup, log_stacked_height, k_whir, log_final_poly_len, security_bits, query_phase_pow_bits,
                     proximity=("unique",)):
    """WhirConfig::new (config.rs:286-327) with the proximity strategy (config.rs:199-277): `("unique",)`,
    `("list", m)` or `("split", m, list_start_round)`."""
    protocol = max(security_bits - query_phase_pow_bits, 0)
    num_rounds = -(-max(log_stacked_height - log_final_poly_len, 0) // k_whir)
    out, log_inv_rate = [], log_blowup
    for rnd in range(num_rounds):
        rho = 2.0 ** (-log_inv_rate)
        if proximity[0] == "unique" or (proximity[0] == "split" and rnd < proximity[2]):
            agree = (1.0 + rho) / 2.0
        else:
            m = float(max(proximity[1], 1))
            agree = math.sqrt(rho) * (1.0 + 1.0 / (2.0 * m))
        agree = min(max(agree, 2.2250738585072014e-308), 1.0)
        out.append(math.ceil(protocol / -math.log2(agree)))
        log_inv_rate += k_whir - 1
    return out


def test_system_params_small(l_skip, n_stack, k_whir, log_final_poly_len=None, max_constraint_degree=3):
    """test_utils/mod.rs:531-579."""
    if log_final_poly_len is None:
        log_final_poly_len = (n_stack + l_skip) % k_whir
    assert log_final_poly_len < l_skip + n_stack
    log_blowup = 1
    return dict(l_skip=l_skip, n_stack=n_stack, w_stack=1 << 12, log_blowup=log_blowup, k_whir=k_whir,
                mu_pow_bits=3, query_phase_pow_bits=1, folding_pow_bits=2, logup_pow_bits=2,
                max_constraint_degree=max_constraint_degree,
                whir_num_queries=whir_num_queries(log_blowup, l_skip + n_stack, k_whir, log_final_poly_len, 5, 1,
                                                  ("split", 3, 1)))


test_system_params_small.__test__ = False  # (a restated reference helper, not a pytest case)


def default_test_params_small():
    """test_utils/mod.rs:581-583."""
    return test_system_params_small(2, 8, 3)


def fib_roundtrip_params(l_skip):
    """fib_air_roundtrip's own parameters (backend-tests/src/lib.rs:262-282): n_stack 8, w_stack 8, k_whir 4,
    log_final_poly_len 4, pow bits 1 / 11 / 10, unique decoding at 80 bits, LogUp pow 18
    (log_up_params.rs:8-9: the 100-bit calibration never goes below 18)."""
    n_stack, k_whir = 8, 4
    return dict(l_skip=l_skip, n_stack=n_stack, w_stack=8, log_blowup=1, k_whir=k_whir, mu_pow_bits=10,
                query_phase_pow_bits=1, folding_pow_bits=11, logup_pow_bits=18, max_constraint_degree=3,
                whir_num_queries=whir_num_queries(1, l_skip + n_stack, k_whir, k_whir, 80, 1))


# ---- keys ---------------------------------------------------------------------------------------------------------
def engine_keygen(air_list):
    """StarkEngine::keygen: every AIR optional."""
    out = []
    for a in air_list:
        a = dict(a)
        a["is_required"] = False
        out.append(a)
    return out


def from_row_major(vals, width):
    m = np.array([int(v) % P for v in vals], dtype=np.uint32).reshape(-1, width)
    return np.ascontiguousarray(m.T).reshape(-1), m.shape[0]


# ---- AIRs and fixtures not already in tests/airs.py -----------------------------------------------------------------
def self_interaction_air(width, bus_index):
    """dummy_airs/interaction/self_interaction_air.rs:26-61, statement by statement -- including that `next_sum` is
    folded over `local` (line 40), so the 'next' interactions carry the LOCAL row's sum as their count."""
    a = Air("SelfInteractionAir", width)
    local = [a.local(i) for i in range(width)]
    nxt = [a.next(i) for i in range(width)]

    def fold_sum(vals):
        acc = Expr.lift(0)
        for v in vals:
            acc = acc + v
        return acc

    local_sum, next_sum = fold_sum(local), fold_sum(local)
    a.push_interaction(bus_index, list(local), ONE(), 1)
    a.push_interaction(bus_index, list(nxt), NEG_ONE(), 1)
    a.push_interaction(bus_index, list(local), local_sum, 1)
    a.push_interaction(bus_index, list(nxt), -next_sum, 1)
    a.push_interaction(bus_index, list(local), local[0], 1)
    a.push_interaction(bus_index, list(nxt), -nxt[0], 1)
    a.push_interaction(bus_index, list(reversed(local)), local_sum, 1)
    a.push_interaction(bus_index, list(reversed(nxt)), -next_sum, 1)
    return a.finish()


def self_interaction_trace(width, log_height):
    """SelfInteractionChip::generate_proving_ctx (:69-84): cell (row, i) = (row + i) % width."""
    h = 1 << log_height
    rows = np.arange(h, dtype=np.int64).reshape(-1, 1)
    m = ((rows + np.arange(width, dtype=np.int64)) % width).astype(np.uint32)
    return np.ascontiguousarray(m.T).reshape(-1), h


def self_interaction_fixture(widths, log_height, bus_index):
    """SelfInteractionFixture (test_utils/mod.rs:402-435)."""
    air_list = [self_interaction_air(w, bus_index) for w in widths]
    traces = []
    for i, w in enumerate(widths):
        tr, h = self_interaction_trace(w, log_height)
        traces.append((i, tr, h, []))
    return air_list, traces


def fib_fixture(a, b, n, num_airs=1, empty_air_indices=()):
    """FibFixture (test_utils/mod.rs:152-219)."""
    tr, pvs = airs.fibonacci_trace(a, b, n)
    return ([airs.fibonacci_air()] * num_airs,
            [(i, tr, n, pvs) for i in range(num_airs) if i not in empty_air_indices])


def interactions_fixture_11():
    """InteractionsFixture11 (test_utils/mod.rs:221-256)."""
    s, hs = from_row_major([0, 1, 3, 5, 7, 4, 546, 889], 2)
    r, hr = from_row_major([1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 889, 1, 889, 0, 456], 2)
    return ([airs.dummy_interaction_air(1, True, 0), airs.dummy_interaction_air(1, False, 0)],
            [(0, s, hs, []), (1, r, hr, [])])


def preprocessed_fib_fixture(a, b, sels):
    """PreprocessedFibFixture (test_utils/mod.rs:333-353); returns (airs, traces, preprocessed)."""
    tr, pvs = airs.fibonacci_selector_trace(a, b, sels)
    pre = {0: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    return [airs.fibonacci_selector_air()], [(0, tr, len(sels), pvs)], pre


def preprocessed_and_cached_fixture(sels, num_cached_parts):
    """PreprocessedAndCachedFixture (test_utils/mod.rs:355-400)."""
    common, cached = airs.preprocessed_cached_traces(sels, num_cached_parts)
    pre = {0: (np.array([int(s) for s in sels], np.uint32), len(sels))}
    return [airs.preprocessed_cached_air(num_cached_parts)], [(0, common, len(sels), [], cached)], pre


def dummy_interaction_chip(field_width, is_send, bus, count, fields, partition=False):
    """DummyInteractionChip::{load_data, generate_proving_ctx} (dummy_interaction_air.rs:138-300): rows padded with
    zeros to the next power of two; with `partition` the count is the common main and the fields are cached main 0.
    Returns (air, trace tuple without its air index)."""
    h = len(count)
    assert len(fields) == h and all(len(f) == field_width for f in fields)
    hp = 1 << max(h - 1, 0).bit_length()
    cnt = np.zeros(hp, np.uint32)
    cnt[:h] = count
    fl = np.zeros((hp, field_width), np.uint32)
    fl[:h] = np.array(fields, dtype=np.uint32).reshape(h, field_width)
    air = airs.dummy_interaction_air(field_width, is_send, bus, partition=partition)
    if partition:
        return air, (cnt, hp, [], np.ascontiguousarray(fl.T).reshape(-1))
    return air, (np.concatenate([cnt, np.ascontiguousarray(fl.T).reshape(-1)]), hp, [])


def _merge(parts):
    """Concatenate fixtures [(airs, traces, preprocessed or None)] the way MixtureFixture does (test_utils/mod.rs:
    474-497): AIR lists flattened, contexts enumerated in that order."""
    air_list, traces, pre = [], [], {}
    for part in parts:
        a, t = part[0], part[1]
        p = part[2] if len(part) > 2 and part[2] else {}
        base = len(air_list)
        air_list += a
        traces += [(x[0] + base,) + tuple(x[1:]) for x in t]
        pre.update({k + base: v for k, v in p.items()})
    return air_list, traces, pre


def mixture_fixture_standard(log_height):
    """MixtureFixture::standard (test_utils/mod.rs:459-472)."""
    height = 1 << log_height
    sels = [i % 2 == 0 for i in range(height)]
    c_airs, c_traces = airs.cached_fixture_11()
    return _merge([fib_fixture(8, 8, height), interactions_fixture_11(), (c_airs, c_traces),
                   preprocessed_fib_fixture(7, 3, sels), self_interaction_fixture([4, 7, 8, 8, 10, 100], log_height, 5)])


# ---- the cases ------------------------------------------------------------------------------------------------------
def _case(params, air_list, traces, pre=None, expect="ok"):
    return Case(params, engine_keygen(air_list), traces, pre or None, expect)


def interactions_single_sender_receiver_happy():
    """lib.rs:166-173"""
    return _case(default_test_params_small(), *interactions_fixture_11())


def single_cached_trace_stark():
    """lib.rs:175-182"""
    return _case(default_test_params_small(), *airs.cached_fixture_11())


def single_preprocessed_trace_stark(log_trace_degree):
    """lib.rs:184-196 (suite: 10, 3, 2, 1, 0)"""
    sels = [i % 2 == 0 for i in range(1 << log_trace_degree)]
    return _case(default_test_params_small(), *preprocessed_fib_fixture(0, 1, sels))


def multi_interaction_traces_stark(log_trace_degree):
    """lib.rs:198-211 (suite: 10, 3, 2, 1, 0)"""
    return _case(default_test_params_small(), *self_interaction_fixture([4, 7, 8, 8, 10, 100], log_trace_degree, 4))


def mixture_traces_stark(log_trace_degree):
    """lib.rs:213-220 (suite: 10, 3, 2, 1, 0)"""
    return _case(default_test_params_small(), *mixture_fixture_standard(log_trace_degree))


def matrix_stacking_overflow():
    """lib.rs:222-235: 37 lifted columns of 2 rows in a stacked matrix 2^8 rows tall -- they spill into a second
    stacked column."""
    return _case(test_system_params_small(3, 5, 3), *self_interaction_fixture([4, 7, 8, 8, 10], 1, 4))


def fib_air_roundtrip(l_skip, log_trace_degree):
    """lib.rs:241-284 (suite: (2, 10), (2, 1), (2, 0), (3, 2), (6, 10))"""
    return _case(fib_roundtrip_params(l_skip), *fib_fixture(0, 1, 1 << log_trace_degree))


def dummy_interactions_roundtrip(l_skip, n_stack, k_whir):
    """lib.rs:286-302 (suite: (2, 8, 3), (5, 5, 4))"""
    return _case(test_system_params_small(l_skip, n_stack, k_whir), *interactions_fixture_11())


def cached_trace_roundtrip(l_skip, n_stack, k_whir):
    """lib.rs:304-321 (suite: (2, 8, 3), (5, 5, 4), (5, 8, 3), (6, 7, 3))"""
    return _case(test_system_params_small(l_skip, n_stack, k_whir), *airs.cached_fixture_11())


def preprocessed_trace_roundtrip(l_skip, n_stack, k_whir):
    """lib.rs:323-343 (suite: (2, 8, 3), (5, 5, 4))"""
    sels = [i % 2 == 0 for i in range(1 << 8)]
    return _case(test_system_params_small(l_skip, n_stack, k_whir), *preprocessed_fib_fixture(0, 1, sels))


def preprocessed_and_cached_trace_roundtrip(l_skip, n_stack, k_whir, num_cached_parts):
    """lib.rs:345-366 (suite: (2, 8, 3, 1..3), (5, 5, 4, 1..3))"""
    sels = [i % 2 == 0 for i in range(1 << 8)]
    return _case(test_system_params_small(l_skip, n_stack, k_whir), *preprocessed_and_cached_fixture(sels, num_cached_parts))


def single_fib_and_dummy_trace_stark(log_trace_degree):
    """lib.rs:571-640 (suite: 3, 2, 1, 0): InteractionsFixture11's AIRs with its traces repeated to 16 and 32 rows,
    plus a Fibonacci AIR of 2 * 2^log_trace_degree rows."""
    sender_height = 2 * (1 << 3)
    cyc = lambda pat, n: [pat[i % len(pat)] for i in range(n)]
    s, hs = from_row_major(cyc([0, 1, 3, 5, 7, 4, 546, 889], 2 * sender_height), 2)
    r, hr = from_row_major(cyc([1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 889, 1, 889, 0, 456], 4 * sender_height), 2)
    fx_airs, _ = interactions_fixture_11()
    f_airs, f_traces = fib_fixture(0, 1, 2 * (1 << log_trace_degree))
    traces = [(0, s, hs, []), (1, r, hr, []), (2,) + tuple(f_traces[0][1:])]
    return _case(default_test_params_small(), fx_airs + f_airs, traces)


def _optional_air_key():
    """lib.rs:649-667: [FibonacciAir, sender, partitioned sender, receiver], one key for the three contexts."""
    return [airs.fibonacci_air(), airs.dummy_interaction_air(1, True, 0),
            airs.dummy_interaction_air(1, True, 0, partition=True), airs.dummy_interaction_air(1, False, 0)]


def optional_air(case):
    """lib.rs:642-750. case 1: every AIR present; case 2: only AIRs 1 and 3 (FibonacciAir and the partitioned sender
    have no trace); case 3: only the receiver -- the prover must fail."""
    key = _optional_air_key()
    f3 = [[1], [2], [3]]
    if case == 1:
        _, fib = fib_fixture(0, 1, 8)
        _, s1 = dummy_interaction_chip(1, True, 0, [1, 2, 4], f3)
        _, s2 = dummy_interaction_chip(1, True, 0, [1, 2, 8], f3, partition=True)
        _, r1 = dummy_interaction_chip(1, False, 0, [2, 4, 12], f3)
        return _case(default_test_params_small(), key, [fib[0], (1,) + s1, (2,) + s2, (3,) + r1])
    if case == 2:
        _, s1 = dummy_interaction_chip(1, True, 0, [1, 2, 4], f3)
        _, r1 = dummy_interaction_chip(1, False, 0, [1, 2, 4], f3)
        return _case(default_test_params_small(), key, [(1,) + s1, (3,) + r1])
    _, r1 = dummy_interaction_chip(1, False, 0, [1, 2, 4], f3)
    return _case(default_test_params_small(), key, [(3,) + r1], expect="prover_err")


def _sender_receiver(sender_rows, receiver_rows, n_senders=1, n_receivers=1, expect="ok"):
    air_list = [airs.dummy_interaction_air(1, True, 0)] * n_senders + [airs.dummy_interaction_air(1, False, 0)] * n_receivers
    traces = []
    for i, vals in enumerate(sender_rows + receiver_rows):
        tr, h = from_row_major(vals, 2)
        traces.append((i, tr, h, []))
    return _case(default_test_params_small(), air_list, traces, expect=expect)


def interaction_multi_rows_neg():
    """lib.rs:844-880: the receiver takes 545 of a field the sender sends 546 times."""
    return _sender_receiver([[0, 1, 3, 5, 7, 4, 546, 0]], [[1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 0, 0, 0, 0, 456]],
                            expect="prover_err")


def interaction_all_zero_sender():
    """lib.rs:882-903: one sender whose every count is zero, no receiver."""
    return _sender_receiver([[0, 1, 0, 5, 0, 4, 0, 889]], [], n_receivers=0)


_RECEIVER = [1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 889, 1, 889, 0, 456]


def interaction_multi_senders():
    """lib.rs:905-939"""
    return _sender_receiver([[0, 1, 3, 5, 6, 4, 333, 889], [1, 4, 213, 889]], [_RECEIVER], n_senders=2)


def interaction_multi_senders_neg():
    """lib.rs:941-980: 6 -> 5 in the first sender."""
    return _sender_receiver([[0, 1, 3, 5, 5, 4, 333, 889], [1, 4, 213, 889]], [_RECEIVER], n_senders=2, expect="prover_err")


def interaction_multi_sender_receiver():
    """lib.rs:982-1018"""
    return _sender_receiver([[0, 1, 3, 5, 6, 4, 333, 889], [1, 4, 213, 889]],
                            [[1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 889, 0, 289, 0, 456], [1, 889]], n_senders=2, n_receivers=2)


def interaction_cached_trace_neg():
    """lib.rs:1020-1074: a plain sender and a partitioned receiver of 2-field messages that do not balance; the AIR
    order is [receiver, sender]. The reference accepts either a prover failure or a proof the verifier rejects; with a
    non-zero root sum it is the prover that fails."""
    s_air, s = dummy_interaction_chip(2, True, 0, [0, 7, 3, 546], [[1, 1], [4, 2], [5, 1], [889, 4]])
    r_air, r = dummy_interaction_chip(2, False, 0, [1, 3, 4, 2, 0, 545, 1, 0],
                                      [[5, 1], [4, 2], [4, 2], [5, 1], [123, 3], [889, 4], [889, 10], [456, 5]], partition=True)
    return _case(default_test_params_small(), [r_air, s_air], [(0,) + r, (1,) + s], expect="prover_err")


def suite(max_log_height=10):
    """[(name, case factory)] in the order of `backend_test_suite!`; heights above `max_log_height` are left out."""
    out = [("interactions_single_sender_receiver_happy", interactions_single_sender_receiver_happy),
           ("single_cached_trace_stark", single_cached_trace_stark)]
    for lg in (10, 3, 2, 1, 0):
        if lg <= max_log_height:
            out.append((f"single_preprocessed_trace_stark[{lg}]", lambda lg=lg: single_preprocessed_trace_stark(lg)))
    for lg in (10, 3, 2, 1, 0):
        if lg <= max_log_height:
            out.append((f"multi_interaction_traces_stark[{lg}]", lambda lg=lg: multi_interaction_traces_stark(lg)))
    for lg in (10, 3, 2, 1, 0):
        if lg <= max_log_height:
            out.append((f"mixture_traces_stark[{lg}]", lambda lg=lg: mixture_traces_stark(lg)))
    out.append(("matrix_stacking_overflow", matrix_stacking_overflow))
    for l_skip, lg in ((2, 10), (2, 1), (2, 0), (3, 2), (6, 10)):
        if lg <= max_log_height:
            out.append((f"fib_air_roundtrip[{l_skip}-{lg}]", lambda a=l_skip, b=lg: fib_air_roundtrip(a, b)))
    for p in ((2, 8, 3), (5, 5, 4)):
        out.append((f"dummy_interactions_roundtrip[{p[0]}-{p[1]}-{p[2]}]", lambda p=p: dummy_interactions_roundtrip(*p)))
    for p in ((2, 8, 3), (5, 5, 4), (5, 8, 3), (6, 7, 3)):
        out.append((f"cached_trace_roundtrip[{p[0]}-{p[1]}-{p[2]}]", lambda p=p: cached_trace_roundtrip(*p)))
    for p in ((2, 8, 3), (5, 5, 4)):
        out.append((f"preprocessed_trace_roundtrip[{p[0]}-{p[1]}-{p[2]}]", lambda p=p: preprocessed_trace_roundtrip(*p)))
    for p in ((2, 8, 3), (5, 5, 4)):
        for parts in (1, 2, 3):
            out.append((f"preprocessed_and_cached_trace_roundtrip[{p[0]}-{p[1]}-{p[2]}-{parts}]",
                        lambda p=p, n=parts: preprocessed_and_cached_trace_roundtrip(*p, n)))
    for lg in (3, 2, 1, 0):
        out.append((f"single_fib_and_dummy_trace_stark[{lg}]", lambda lg=lg: single_fib_and_dummy_trace_stark(lg)))
    for c in (1, 2, 3):
        out.append((f"optional_air[case{c}]", lambda c=c: optional_air(c)))
    out += [("interaction_multi_rows_neg", interaction_multi_rows_neg),
            ("interaction_all_zero_sender", interaction_all_zero_sender),
            ("interaction_multi_senders", interaction_multi_senders),
            ("interaction_multi_senders_neg", interaction_multi_senders_neg),
            ("interaction_multi_sender_receiver", interaction_multi_sender_receiver),
            ("interaction_cached_trace_neg", interaction_cached_trace_neg)]
    return out
