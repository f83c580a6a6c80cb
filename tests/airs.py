"""Test AIRs and the proving-key blob (test infrastructure).

Restates, in Python, the reference's symbolic pipeline just far enough to produce the inputs of
the prove hot path for the reference's own fixture AIRs:
  * SymbolicExpression ops and degree tracking  crates/stark-backend/src/air_builders/symbolic/symbolic_expression.rs:22-330
  * DAG builder with dedup + simplifications    crates/stark-backend/src/air_builders/symbolic/dag.rs:95-330
  * FibonacciAir                                crates/stark-backend/src/test_utils/dummy_airs/fib_air/air.rs:26-52
  * DummyInteractionAir (sender / receiver)     .../dummy_airs/interaction/dummy_interaction_air.rs:101-131
  * BenchmarkAir (uniform_runner)               benchmarks/synthetic/src/bin/uniform_runner.rs:79-113
  * production / test SystemParams              crates/stark-sdk/src/config/mod.rs:121-138, crates/stark-backend/src/config.rs:286-353
The blob is the flat u32 encoding both the oracle C API and the product C ABI accept
(include/openvm_hip.h, "proving-key blob").
"""
import math

import numpy as np

P = 2013265921

N_VAR, N_IS_FIRST, N_IS_LAST, N_IS_TRANSITION, N_CONST, N_ADD, N_SUB, N_NEG, N_MUL = range(9)
E_PREP, E_MAIN, E_PUBLIC = 0, 1, 2


class Expr:
    """SymbolicExpression: no folding at this level, only degree tracking."""

    def __init__(self, kind, deg, **kw):
        self.kind, self.deg = kind, deg
        self.__dict__.update(kw)

    @staticmethod
    def lift(x):
        return x if isinstance(x, Expr) else Expr(N_CONST, 0, value=int(x) % P)

    def __add__(self, o):
        o = Expr.lift(o)
        return Expr(N_ADD, max(self.deg, o.deg), x=self, y=o)

    def __sub__(self, o):
        o = Expr.lift(o)
        return Expr(N_SUB, max(self.deg, o.deg), x=self, y=o)

    def __mul__(self, o):
        o = Expr.lift(o)
        return Expr(N_MUL, self.deg + o.deg, x=self, y=o)

    def __neg__(self):
        return Expr(N_NEG, self.deg, x=self)

    __radd__ = lambda self, o: Expr.lift(o) + self
    __rsub__ = lambda self, o: Expr.lift(o) - self
    __rmul__ = lambda self, o: Expr.lift(o) * self


def var(entry, index, part_index=0, offset=0):
    return Expr(N_VAR, 1 if entry != E_PUBLIC else 0, entry=entry, index=index,
                part_index=part_index, offset=offset)


IS_FIRST = lambda: Expr(N_IS_FIRST, 1)
IS_LAST = lambda: Expr(N_IS_LAST, 1)
IS_TRANSITION = lambda: Expr(N_IS_TRANSITION, 1)
ONE = lambda: Expr(N_CONST, 0, value=1)
NEG_ONE = lambda: Expr(N_CONST, 0, value=P - 1)


class DagBuilder:
    """dag.rs:143-330"""

    def __init__(self):
        self.expr_to_idx, self.node_to_idx, self.nodes = {}, {}, []

    def intern(self, node):
        if node not in self.node_to_idx:
            self.node_to_idx[node] = len(self.nodes)
            self.nodes.append(node)
        return self.node_to_idx[node]

    def const_of(self, idx):
        n = self.nodes[idx]
        return n[1] if n[0] == N_CONST else None

    def neg_child(self, idx):
        n = self.nodes[idx]
        return n[1] if n[0] == N_NEG else None

    def add(self, e):
        if id(e) in self.expr_to_idx:
            return self.expr_to_idx[id(e)]
        k = e.kind
        if k == N_VAR:
            idx = self.intern((N_VAR, e.entry, e.part_index, e.offset, e.index))
        elif k in (N_IS_FIRST, N_IS_LAST, N_IS_TRANSITION):
            idx = self.intern((k,))
        elif k == N_CONST:
            idx = self.intern((N_CONST, e.value))
        elif k == N_NEG:
            c = self.add(e.x)
            cv = self.const_of(c)
            idx = self.intern((N_CONST, (-cv) % P)) if cv is not None else self.intern((N_NEG, c))
        else:
            l, r = self.add(e.x), self.add(e.y)
            a, b = self.const_of(l), self.const_of(r)
            if k == N_ADD:
                if a is not None and b is not None:
                    idx = self.intern((N_CONST, (a + b) % P))
                elif a == 0:
                    idx = r
                elif b == 0:
                    idx = l
                elif self.neg_child(r) is not None:
                    idx = self.intern((N_SUB, l, self.neg_child(r)))
                else:
                    idx = self.intern((N_ADD, l, r))
            elif k == N_SUB:
                if a is not None and b is not None:
                    idx = self.intern((N_CONST, (a - b) % P))
                elif b == 0:
                    idx = l
                elif self.neg_child(r) is not None:
                    idx = self.intern((N_ADD, l, self.neg_child(r)))
                else:
                    idx = self.intern((N_SUB, l, r))
            else:
                if a is not None and b is not None:
                    idx = self.intern((N_CONST, a * b % P))
                elif a == 0 or b == 1:
                    idx = l
                elif b == 0 or a == 1:
                    idx = r
                else:
                    idx = self.intern((N_MUL, l, r))
        self.expr_to_idx[id(e)] = idx
        self._keep = getattr(self, "_keep", [])
        self._keep.append(e)  # keep ids unique for the builder's lifetime
        return idx


class Air:
    """Collects constraints / interactions like SymbolicRapBuilder (symbolic/mod.rs:99-258)."""

    def __init__(self, name, common_main_width, num_public_values=0, is_required=True, preprocessed_width=0,
                 cached_main_widths=()):
        self.name, self.width, self.num_pvs, self.is_required = name, common_main_width, num_public_values, is_required
        self.preprocessed_width, self.cached_widths = preprocessed_width, list(cached_main_widths)
        self.constraints, self.interactions = [], []

    # AirBuilder surface. Main partitions: cached_0 .. cached_{m-1}, then the common main as part m
    # (symbolic/mod.rs:117-127).
    def local(self, i):
        return var(E_MAIN, i, len(self.cached_widths), 0)

    def next(self, i):
        return var(E_MAIN, i, len(self.cached_widths), 1)

    def cached(self, part, i, offset=0):
        return var(E_MAIN, i, part, offset)

    def preprocessed(self, i, offset=0):
        return var(E_PREP, i, 0, offset)

    def public(self, i):
        return var(E_PUBLIC, i)

    def assert_zero(self, x, when=None):
        x = Expr.lift(x)
        self.constraints.append(when * x if when is not None else x)

    def assert_eq(self, x, y, when=None):
        self.assert_zero(Expr.lift(x) - Expr.lift(y), when)

    def assert_bool(self, x):
        self.assert_zero(x * (x - ONE()))  # p3-field bool_check

    def push_interaction(self, bus, fields, count, count_weight=0):
        self.interactions.append((bus, [Expr.lift(f) for f in fields], Expr.lift(count), count_weight))

    def finish(self):
        """build_symbolic_constraints_dag (dag.rs:95-129) + vk facts (keygen/mod.rs:262-310)."""
        b = DagBuilder()
        cidx = sorted(set(b.add(c) for c in self.constraints))
        inter = []
        for bus, fields, count, cw in self.interactions:
            msg = [b.add(f) for f in fields]
            cnt = b.add(count)
            inter.append((bus, cw, cnt, msg))
        degs = [c.deg for c in self.constraints]
        for _, fields, count, _ in self.interactions:
            degs += [count.deg] + [f.deg for f in fields]
        need_rot = any(n[0] == N_VAR and n[1] != E_PUBLIC and n[3] > 0 for n in b.nodes)
        return dict(name=self.name, nodes=list(b.nodes), constraint_idx=cidx, interactions=inter,
                    need_rot=need_rot, is_required=self.is_required,
                    max_constraint_degree=max(degs) if degs else 0, num_public_values=self.num_pvs,
                    common_main_width=self.width, preprocessed_width=self.preprocessed_width,
                    cached_main_widths=list(self.cached_widths))


# ---------------------------------------------------------------------------------------------
def fibonacci_air():
    a = Air("FibonacciAir", 2, num_public_values=3)
    pa, pb, px = a.public(0), a.public(1), a.public(2)
    a.assert_eq(a.local(0), pa, when=IS_FIRST())
    a.assert_eq(a.local(1), pb, when=IS_FIRST())
    a.assert_eq(a.local(1), a.next(0), when=IS_TRANSITION())
    a.assert_eq(a.local(0) + a.local(1), a.next(1), when=IS_TRANSITION())
    a.assert_eq(a.local(1), px, when=IS_LAST())
    return a.finish()


def fibonacci_trace(a, b, n):
    rows = [[a % P, b % P]]
    for _ in range(1, n):
        rows.append([rows[-1][1], (rows[-1][0] + rows[-1][1]) % P])
    m = np.array(rows, dtype=np.uint32)          # n x 2
    return np.ascontiguousarray(m.T).reshape(-1), [a % P, b % P, int(m[-1][1])]


def dummy_interaction_air(field_width, is_send, bus, partition=False):
    """dummy_interaction_air.rs:94-121; `partition`: count in the common main, fields in cached main 0."""
    if partition:
        a = Air("DummyInteractionAir", 1, cached_main_widths=[field_width])
        count = a.local(0)
        fields = [a.cached(0, i) for i in range(field_width)]
    else:
        a = Air("DummyInteractionAir", 1 + field_width)
        count = a.local(0)
        fields = [a.local(1 + i) for i in range(field_width)]
    if is_send:
        a.push_interaction(bus, fields, count)
    else:
        a.push_interaction(bus, fields, NEG_ONE() * count)
    return a.finish()


def benchmark_air(num_columns, total_constraints, total_interaction_pairs):
    a = Air("BenchmarkAir", num_columns)
    for i in range(total_constraints):
        a.assert_bool(a.local(i % num_columns))
    for i in range(total_interaction_pairs):
        a.push_interaction(0, [a.local(i % num_columns)], ONE())
        a.push_interaction(0, [a.local(i % num_columns)], NEG_ONE())
    return a.finish()


def benchmark_air_partitioned(num_columns, total_constraints, total_interaction_pairs):
    """BenchmarkAir with its columns declared as ONE cached-main partition plus a 1-column common main
    (the per-AIR-commitment form of the multi-GPU layout; pattern of
    test_utils/dummy_airs/preprocessed_cached_air/air.rs:22-30). Same constraints and interactions on
    the cached columns; the common column carries one boolean constraint."""
    a = Air("BenchmarkAirPartitioned", 1, cached_main_widths=[num_columns])
    for i in range(total_constraints):
        a.assert_bool(a.cached(0, i % num_columns))
    a.assert_bool(a.local(0))
    for i in range(total_interaction_pairs):
        a.push_interaction(0, [a.cached(0, i % num_columns)], ONE())
        a.push_interaction(0, [a.cached(0, i % num_columns)], NEG_ONE())
    return a.finish()


def uniform_runner_air_partitioned(cols, constraints_per_col=2.0, interactions_per_col=0.25):
    rnd = lambda x: int(math.floor(x + 0.5))
    return benchmark_air_partitioned(cols, rnd(constraints_per_col * cols), rnd(interactions_per_col * cols / 2.0))


def uniform_runner_air(cols, constraints_per_col=2.0, interactions_per_col=0.25):
    """uniform_runner.rs:196-199"""
    rnd = lambda x: int(math.floor(x + 0.5))  # f64::round (half away from zero)
    return benchmark_air(cols, rnd(constraints_per_col * cols), rnd(interactions_per_col * cols / 2.0))


def splitmix64_bits(seed, n):
    """Seeded boolean trace (SURVEY §8d config D value distribution R): the bits, least significant first, of the
    splitmix64 stream from `seed`. Vectorised (the full-size sharded comparison draws 8.5e8 of them)."""
    n64 = (n + 63) // 64
    with np.errstate(over="ignore"):
        x = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.arange(1, n64 + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    bits = np.unpackbits(z.view(np.uint8).reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
    return bits[:n].astype(np.uint32)


# ---------------------------------------------------------------------------------------------
def whir_num_queries(log_blowup, log_stacked_height, k_whir, log_final_poly_len, security_bits,
                     query_phase_pow_bits):
    """WhirConfig::new, unique-decoding regime (config.rs:286-353)."""
    protocol = max(security_bits - query_phase_pow_bits, 0)
    num_rounds = -(-max(log_stacked_height - log_final_poly_len, 0) // k_whir)
    out, log_inv_rate = [], log_blowup
    for _ in range(num_rounds):
        rho = 2.0 ** (-log_inv_rate)
        per_query = -math.log2((1.0 + rho) / 2.0)
        out.append(math.ceil(protocol / per_query))
        log_inv_rate += k_whir - 1
    return out


def make_params(l_skip, n_stack, log_blowup, k_whir, log_final_poly_len, security_bits,
                mu_pow_bits, folding_pow_bits, query_phase_pow_bits, logup_pow_bits,
                max_constraint_degree, w_stack):
    return dict(l_skip=l_skip, n_stack=n_stack, w_stack=w_stack, log_blowup=log_blowup, k_whir=k_whir,
                mu_pow_bits=mu_pow_bits, query_phase_pow_bits=query_phase_pow_bits,
                folding_pow_bits=folding_pow_bits, logup_pow_bits=logup_pow_bits,
                max_constraint_degree=max_constraint_degree,
                whir_num_queries=whir_num_queries(log_blowup, l_skip + n_stack, k_whir,
                                                  log_final_poly_len, security_bits, query_phase_pow_bits))


def app_params(log_stacked_height):
    """app_params_with_100_bits_security (stark-sdk/src/config/mod.rs:121-138): l_skip 4,
    log_blowup 1, k_whir 4, w_stack 2048, pow bits 5/15/20, logup pow 18, max degree 3."""
    return make_params(4, max(log_stacked_height - 4, 0), 1, 4, 10, 100, 15, 5, 20, 18, 3, 2048)


def test_params(l_skip=2, n_stack=8, k_whir=3, log_blowup=1, max_constraint_degree=3):
    """Toy parameters in the spirit of test_utils/mod.rs:517-583 (security 5 bits, pow 1/2/3,
    logup pow 2)."""
    return make_params(l_skip, n_stack, log_blowup, k_whir, 2, 5, 1, 2, 3, 2, max_constraint_degree, 4096)


MAGIC = 0x4F564D31


PREP_MAGIC = 0x50524550


def pk_blob(params, airs, vk_pre_hash=None, preprocessed=None):
    """Flat u32 proving-key blob (field constants canonical). `preprocessed`: {air position:
    (canonical column-major values, height)} for the AIRs with a preprocessed trace."""
    w = [MAGIC, params["l_skip"], params["n_stack"], params["w_stack"], params["log_blowup"],
         params["k_whir"], params["mu_pow_bits"], params["query_phase_pow_bits"],
         params["folding_pow_bits"], params["logup_pow_bits"], params["max_constraint_degree"],
         len(params["whir_num_queries"])] + list(params["whir_num_queries"])
    w += list(vk_pre_hash) if vk_pre_hash is not None else [(7 * i + 1) % P for i in range(8)]
    w.append(len(airs))
    for a in airs:
        cw = a.get("cached_main_widths", [])
        pw = a.get("preprocessed_width", 0)
        w += [int(a["need_rot"]), int(a["is_required"]), a["max_constraint_degree"],
              a["num_public_values"], int(pw > 0), pw, len(cw)] + list(cw) + [a["common_main_width"], len(a["nodes"])]
        for n in a["nodes"]:
            k = n[0]
            if k == N_VAR:
                w += [k, n[1], n[2], n[3], n[4], 0]
            elif k == N_CONST:
                w += [k, 0, 0, 0, n[1], 0]
            elif k == N_NEG:
                w += [k, 0, 0, 0, n[1], 0]
            elif k in (N_ADD, N_SUB, N_MUL):
                w += [k, 0, 0, 0, n[1], n[2]]
            else:
                w += [k, 0, 0, 0, 0, 0]
        w.append(len(a["constraint_idx"]))
        w += a["constraint_idx"]
        w.append(len(a["interactions"]))
        for bus, cw, cnt, msg in a["interactions"]:
            w += [bus, cw, cnt, len(msg)] + msg
    if any(a.get("preprocessed_width", 0) for a in airs):
        w.append(PREP_MAGIC)
        for i, a in enumerate(airs):
            if a.get("preprocessed_width", 0):
                vals, h = preprocessed[i]
                assert len(vals) == a["preprocessed_width"] * h
                w.append(h)
                w += [int(x) for x in vals]
    return np.array(w, dtype=np.uint32)


# ---- fixtures with preprocessed / cached traces (test_utils/mod.rs:261-436) -------------------
def fibonacci_selector_air():
    """fib_selector_air/air.rs:55-113 (interactions disabled): the selector is a 1-column
    preprocessed trace."""
    a = Air("FibonacciSelectorAir", 2, num_public_values=3, preprocessed_width=1)
    pa, pb, px = a.public(0), a.public(1), a.public(2)
    sel = a.preprocessed(0)
    left, right, nleft, nright = a.local(0), a.local(1), a.next(0), a.next(1)
    a.assert_eq(left, pa, when=IS_FIRST())
    a.assert_eq(right, pb, when=IS_FIRST())
    a.assert_eq(right, nleft, when=IS_TRANSITION() * sel)
    a.assert_eq(left, nleft, when=IS_TRANSITION() * (sel - ONE()))
    a.assert_eq(left + right, nright, when=IS_TRANSITION() * sel)
    a.assert_eq(right, nright, when=IS_TRANSITION() * (sel - ONE()))
    a.assert_eq(right, px, when=IS_LAST())
    return a.finish()


def fibonacci_selector_trace(a, b, sels):
    """fib_selector_air/trace.rs:7-22 + get_conditional_fib_number; returns (col-major trace, pvs)."""
    rows = [[a % P, b % P]]
    for i in range(1, len(sels)):
        l, r = rows[-1]
        rows.append([r, (l + r) % P] if sels[i - 1] else [l, r])
    tr = np.array(rows, dtype=np.uint32).T.reshape(-1).copy()
    return tr, [a % P, b % P, rows[-1][1]]


def preprocessed_cached_air(num_cached_parts):
    """preprocessed_cached_air/air.rs:49-120: 1 preprocessed selector column, 1 common column,
    `num_cached_parts` cached single-column partitions."""
    a = Air("PreprocessedCachedAir", 1, preprocessed_width=1, cached_main_widths=[1] * num_cached_parts)
    pl, pn = a.preprocessed(0), a.preprocessed(0, 1)
    cl, cn = a.local(0), a.next(0)
    a.assert_zero(pl * (pl - ONE()))
    a.assert_zero(cl, when=IS_FIRST())
    a.assert_zero((cn - cl) - ONE(), when=IS_TRANSITION())  # assert_one
    prev_l, prev_n = cl, cn
    for part in range(num_cached_parts):
        kl, kn = a.cached(part, 0), a.cached(part, 0, 1)
        a.assert_eq(prev_l + pl, kl)
        a.assert_eq(prev_n + pn, kn, when=IS_TRANSITION())
        prev_l, prev_n = kl, kn
    return a.finish()


def preprocessed_cached_traces(sels, num_cached_parts):
    """test_utils/mod.rs:373-400: common = 0..n; cached part p = i + (p+1)*sel[i]."""
    n = len(sels)
    common = np.arange(n, dtype=np.uint32)
    cached = np.concatenate([np.array([(i + (p + 1) * int(sels[i])) % P for i in range(n)], dtype=np.uint32)
                             for p in range(num_cached_parts)])
    return common, cached


def cached_fixture_11():
    """CachedFixture11 (test_utils/mod.rs:265-331): partitioned sender/receiver; returns
    (airs, traces) with traces = (air_idx, common, height, pvs, cached)."""
    airs_ = [dummy_interaction_air(1, True, 0, partition=True), dummy_interaction_air(1, False, 0, partition=True)]
    traces = [(0, np.array([0, 3, 7, 546], np.uint32), 4, [], np.array([1, 5, 4, 889], np.uint32)),
              (1, np.array([1, 3, 4, 2, 0, 545, 1, 0], np.uint32), 8, [],
               np.array([5, 4, 4, 5, 123, 889, 889, 456], np.uint32))]
    return airs_, traces


def stress_params(log_stacked_height, log_blowup=2, max_constraint_degree=4):
    """BASELINE config E: leaf-like production parameters (stark-sdk/src/config/mod.rs:140-168:
    l_skip 4, log_blowup 2, folding pow 4, mu pow 13, degree 4) at a chosen stacked height."""
    return make_params(4, max(log_stacked_height - 4, 0), log_blowup, 4, 10, 100, 13, 4, 20, 18,
                       max_constraint_degree, 2048)


# ---- synthetic replay of a captured OpenVM workload --------------------------
def load_reth_shapes():
    """tests/golden/reth_block_23992138_shapes.json.gz (derived by tools/make_reth_shapes.py)."""
    import gzip
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reth_block_23992138_shapes.json.gz")
    with gzip.open(path, "rb") as f:
        return json.loads(f.read().decode())["segments"]


def synthetic_air(rec):
    """SyntheticAir::from_shape + eval (benchmarks/synthetic/src/synthetic_air.rs:86-232) for one shape
    record [log_height, common_main_width, num_constraints, num_interactions, max_constraint_degree,
    num_distinct_buses, message_lens, count_weights]. Column 0 is the all-zero 'kill column'. Returns
    (air, width)."""
    _, cm_width, n_constraints, n_inter, max_deg, n_buses, msg_lens, cws = rec
    width = max(cm_width, (max(msg_lens) if msg_lens else 0) + 1, 1)
    n_buses = max(n_buses, 1)
    degree = max(max_deg, 1)
    a = Air("SyntheticAir", width)
    slots = 2 * width
    for c in range(n_constraints):
        idx, term = c, None
        for _ in range(degree):
            slot = idx % slots
            idx //= slots
            col = a.local(slot) if slot < width else a.next(slot - width)
            term = col if term is None else term * col
        a.assert_zero(term)
    for i in range(n_inter):
        msg_len = min(msg_lens[i] if i < len(msg_lens) else min(width - 1, 4), width - 1)
        fields = [a.local((j + 1) % width) for j in range(msg_len)]
        count = a.local(0)
        a.push_interaction(i % n_buses, fields, count if i % 2 == 0 else NEG_ONE() * count,
                           cws[i] if i < len(cws) else 0)
    return a.finish(), width


def synthetic_segment(seg, max_log_height):
    """(air list, [(air_idx, log_height, width)]) for one captured segment, heights clamped as
    synthetic_runner.rs:196-203 does (and raised to 2 rows)."""
    air_list, shapes = [], []
    for i, rec in enumerate(seg["airs"]):
        air, width = synthetic_air(rec)
        air_list.append(air)
        shapes.append((i, max(min(rec[0], max_log_height), 1), width))
    return air_list, shapes


# ---- randomised AIRs for differential testing (not from the reference) ---------------------------
def random_air(rng, width, n_constraints, n_pairs, max_degree=3, use_next=True, num_public_values=0):
    """A random AIR: constraints are random expression trees over local / next cells, constants and
    public values (degree <= max_degree); every interaction is sent and received with the same
    fields and count, so LogUp balances on ANY trace. Constraints are NOT satisfied by random traces:
    both provers still emit a proof (no prover-side validity check), and their bytes must agree."""
    a = Air("RandomAir", width, num_public_values=num_public_values)

    def leaf():
        k = int(rng.integers(0, 10))
        if k < 6 or not use_next:
            return a.local(int(rng.integers(0, width)))
        if k < 8:
            return a.next(int(rng.integers(0, width)))
        if k < 9 and num_public_values:
            return a.public(int(rng.integers(0, num_public_values)))
        return Expr.lift(int(rng.integers(0, P)))

    def tree(depth, deg_budget):
        if depth == 0 or deg_budget <= 1 and rng.integers(0, 2):
            return leaf()
        op = int(rng.integers(0, 4))
        if op == 0 and deg_budget >= 2:
            d1 = int(rng.integers(1, deg_budget))
            return tree(depth - 1, d1) * tree(depth - 1, deg_budget - d1)
        if op == 1:
            return tree(depth - 1, deg_budget) + tree(depth - 1, deg_budget)
        if op == 2:
            return tree(depth - 1, deg_budget) - tree(depth - 1, deg_budget)
        return -tree(depth - 1, deg_budget)

    for _ in range(n_constraints):
        e = tree(3, max_degree)
        sel = int(rng.integers(0, 5))
        when = [None, IS_FIRST(), IS_LAST(), IS_TRANSITION(), None][sel]
        if when is not None and e.deg + 1 > max_degree:
            when = None
        a.assert_zero(e, when=when)
    for i in range(n_pairs):
        n_fields = int(rng.integers(1, 4))
        fields = [tree(1, 1) for _ in range(n_fields)]
        count = a.local(int(rng.integers(0, width)))
        bus = int(rng.integers(0, 3))
        a.push_interaction(bus, fields, count)
        a.push_interaction(bus, fields, NEG_ONE() * count)
    return a.finish()


def drop_random_airs(rng, air_list, traces, p_optional=0.5, p_absent=0.5):
    """Marks a random subset of the AIRs optional (`MultiStarkKeygenBuilder::add_air`, keygen/mod.rs:58-60; what
    `StarkEngine::keygen` does for every AIR) and removes the traces of a random subset of those -- the contexts of the
    reference's `optional_air` test (crates/backend-tests/src/lib.rs:642-750). At least one trace stays. Returns
    (air_list, traces, number of absent AIRs); the AIRs are copied, the traces keep their AIR indices."""
    out, absent = [], set()
    for i, a in enumerate(air_list):
        a = dict(a)
        if rng.random() < p_optional:
            a["is_required"] = False
            if rng.random() < p_absent:
                absent.add(i)
        out.append(a)
    if len(absent) == len(air_list):
        absent.discard(int(rng.integers(0, len(air_list))))
    return out, [t for t in traces if t[0] not in absent], len(absent)


# ---- Keccak-f[1600] AIR (BASELINE config B; crates/cpu-backend/benches/keccakf.rs:49-86) ---------------------
# The reference proves p3-keccak-air 0.4.3's `KeccakAir` (a Plonky3 crate that is NOT in /root/reference and cannot be
# fetched here). Restated from the published design of that AIR -- one row per round, 24 rows per permutation:
# theta through the column parities C, C' and the bits of A' (A = A' xor C xor C'), rho/pi as a re-indexing of A' bits,
# chi on 16-bit limbs, iota through the bits of A''[0][0] -- with its column layout (`KeccakCols`, 2 633 columns) and its
# constraint list in its `eval` order (3 182 constraints, degree 3). Cross-checks against facts that ARE in the reference:
# OpenVM's KeccakfPermAir wraps this AIR with one extra column and one extra constraint, and the captured workload shape
# of that chip is 2 634 columns / 3 183 constraints (tests/golden/reth_block_23992138_shapes.json.gz, derived from
# benchmarks/synthetic/reth-block-23992138-profile.jsonl). Constraint ORDER only matters for bit-identity against the
# real crate, which nothing here can check; the trace generator is checked against an independent Keccak-f[1600]
# (hashlib's SHA3 through the sponge) in tests/test_keccak_air.py.
KECCAK_ROUNDS = 24
KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
             0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
             0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
             0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
             0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
KECCAK_R = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]  # r[x][y]


class KeccakCols:
    """Column offsets of `KeccakCols<T>` (repr(C) field order)."""
    STEP_FLAGS = 0                       # [24]
    EXPORT = 24
    PREIMAGE = 25                        # [y][x][limb]  5 x 5 x 4
    A = PREIMAGE + 100                   # [y][x][limb]
    C = A + 100                          # [x][z]        5 x 64
    C_PRIME = C + 320                    # [x][z]
    A_PRIME = C_PRIME + 320              # [y][x][z]     5 x 5 x 64
    A_PRIME_PRIME = A_PRIME + 1600       # [y][x][limb]
    A_PP_00_BITS = A_PRIME_PRIME + 100   # [64]
    A_PPP_00_LIMBS = A_PP_00_BITS + 64   # [4]
    WIDTH = A_PPP_00_LIMBS + 4           # 2633

    @staticmethod
    def b_source(x, y, z):
        """B[x][y] bit z is A' bit: B[x, y] = ROT(A'[a, b], r[a][b]) with a = (x + 3y) % 5, b = x. -> (y, x, z) of a_prime"""
        a, b = (x + 3 * y) % 5, x
        return b, a, (z + 64 - KECCAK_R[a][b]) % 64


def keccak_air():
    K = KeccakCols
    air = Air("KeccakAir", K.WIDTH)
    L, N = air.local, air.next
    TWO = Expr.lift(2)

    def xor(p, q):
        return p + q - p * (q + q)

    def xor3(p, q, r):
        return xor(xor(p, q), r)

    def andn(p, q):
        return (ONE() - p) * q

    def limb_of(bit, limb):
        acc = Expr.lift(0)
        for z in reversed(range(16 * limb, 16 * limb + 16)):
            acc = (acc + acc) + bit(z)
        return acc

    flag = lambda r: L(K.STEP_FLAGS + r)
    pre = lambda y, x, l, row=L: row(K.PREIMAGE + (5 * y + x) * 4 + l)
    a_in = lambda y, x, l, row=L: row(K.A + (5 * y + x) * 4 + l)
    c = lambda x, z: L(K.C + 64 * x + z)
    cp = lambda x, z: L(K.C_PRIME + 64 * x + z)
    ap = lambda y, x, z: L(K.A_PRIME + (5 * y + x) * 64 + z)
    app = lambda y, x, l: L(K.A_PRIME_PRIME + (5 * y + x) * 4 + l)
    app00 = lambda z: L(K.A_PP_00_BITS + z)
    appp = lambda y, x, l: L(K.A_PPP_00_LIMBS + l) if (y, x) == (0, 0) else app(y, x, l)
    b = lambda x, y, z: ap(*K.b_source(x, y, z))

    # round flags (round_flags.rs): first row starts a permutation, the flags rotate
    air.assert_zero(flag(0) - ONE(), when=IS_FIRST())
    for r in range(1, KECCAK_ROUNDS):
        air.assert_zero(flag(r), when=IS_FIRST())
    for r in range(KECCAK_ROUNDS):
        air.assert_zero(flag(r) - N(K.STEP_FLAGS + (r + 1) % KECCAK_ROUNDS), when=IS_TRANSITION())

    first_step, final_step = flag(0), flag(KECCAK_ROUNDS - 1)
    not_final = ONE() - final_step
    # first step: the input A is the preimage
    for y in range(5):
        for x in range(5):
            for l in range(4):
                air.assert_zero(pre(y, x, l) - a_in(y, x, l), when=first_step)
    # not the final step: the preimage is carried to the next row
    for y in range(5):
        for x in range(5):
            for l in range(4):
                air.assert_zero(pre(y, x, l) - pre(y, x, l, N), when=not_final * IS_TRANSITION())
    air.assert_bool(L(K.EXPORT))
    air.assert_zero(L(K.EXPORT), when=not_final)
    # C'[x, z] = xor(C[x, z], C[x - 1, z], C[x + 1, z - 1])
    for x in range(5):
        for z in range(64):
            air.assert_bool(c(x, z))
        for z in range(64):
            air.assert_zero(cp(x, z) - xor3(c(x, z), c((x + 4) % 5, z), c((x + 1) % 5, (z + 63) % 64)))
    # A[x, y, z] = xor(A'[x, y, z], C[x, z], C'[x, z]), checked per 16-bit limb of A
    for y in range(5):
        for x in range(5):
            for z in range(64):
                air.assert_bool(ap(y, x, z))
            for l in range(4):
                air.assert_zero(limb_of(lambda z: xor3(ap(y, x, z), c(x, z), cp(x, z)), l) - a_in(y, x, l))
    # xor_y A'[x, y, z] = C'[x, z]: diff in {0, 2, 4}
    for x in range(5):
        for z in range(64):
            s = ap(0, x, z)
            for y in range(1, 5):
                s = s + ap(y, x, z)
            diff = s - cp(x, z)
            air.assert_zero(diff * (diff - TWO) * (diff - Expr.lift(4)))
    # A''[x, y] = xor(B[x, y], andn(B[x + 1, y], B[x + 2, y]))
    for y in range(5):
        for x in range(5):
            for l in range(4):
                bit = lambda z: xor(andn(b((x + 1) % 5, y, z), b((x + 2) % 5, y, z)), b(x, y, z))
                air.assert_zero(limb_of(bit, l) - app(y, x, l))
    # A'''[0, 0] = A''[0, 0] xor RC
    for z in range(64):
        air.assert_bool(app00(z))
    for l in range(4):
        air.assert_zero(limb_of(app00, l) - app(0, 0, l))

    def xored_bit(z):
        rc = Expr.lift(0)
        for r in range(KECCAK_ROUNDS):
            rc = rc + flag(r) * Expr.lift((KECCAK_RC[r] >> z) & 1)
        return xor(rc, app00(z))
    for l in range(4):
        air.assert_zero(limb_of(xored_bit, l) - L(K.A_PPP_00_LIMBS + l))
    # this round's output is the next round's input
    for x in range(5):
        for y in range(5):
            for l in range(4):
                air.assert_zero(appp(y, x, l) - a_in(y, x, l, N), when=IS_TRANSITION() * not_final)
    assert len(air.constraints) == 3182
    return air.finish()


def keccak_f1600(lanes):
    """Plain Keccak-f[1600] on 25 python ints, lane index x + 5y (FIPS 202) -- the trace generator's cross-check."""
    M = (1 << 64) - 1
    rot = lambda v, n: ((v << n) | (v >> (64 - n))) & M if n else v
    a = list(lanes)
    for rnd in range(KECCAK_ROUNDS):
        cc = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [cc[(x + 4) % 5] ^ rot(cc[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        bb = [0] * 25
        for x in range(5):
            for y in range(5):
                bb[y + 5 * ((2 * x + 3 * y) % 5)] = rot(a[x + 5 * y], KECCAK_R[x][y])
        a = [bb[x + 5 * y] ^ (~bb[(x + 1) % 5 + 5 * y] & bb[(x + 2) % 5 + 5 * y] & M) for y in range(5) for x in range(5)]
        a[0] ^= KECCAK_RC[rnd]
    return a


def keccak_inputs(n_perms, seed=42):
    """Seeded 25-lane inputs as an (n_perms, 25) uint64 array, lane index x + 5y. NOT the reference bench's
    `StdRng::seed_from_u64(42)` stream (ChaCha12 is not restated; any inputs make a valid trace): splitmix64."""
    M = 0xFFFFFFFFFFFFFFFF
    x0 = (seed * 0x9E3779B97F4A7C15 + 1) & M
    with np.errstate(over="ignore"):
        x = np.uint64(x0) + np.arange(1, n_perms * 25 + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        out = z ^ (z >> np.uint64(31))
    return out.reshape(n_perms, 25)


def keccak_trace(inputs, log_rows, device="cpu", mont_one=None):
    """`generate_trace_rows` for KeccakAir: 24 rows per permutation, zero-input permutations appended (the last one cut
    off) up to 2^log_rows rows; `export` stays 0. Vectorised over the permutations in torch (runs on the GPU for the
    2^20-row config): returns the COLUMN-MAJOR trace as a flat int32/int64 tensor of 2633 * 2^log_rows values --
    canonical, or multiplied by `mont_one` (then Montgomery words) when given -- plus the final states (n_perms, 25)
    of the real inputs as python ints for cross-checking."""
    import torch
    K = KeccakCols
    h = 1 << log_rows
    n_real = len(inputs)
    n_perm = -(-h // KECCAK_ROUNDS)
    assert n_real <= n_perm
    lanes = np.zeros((n_perm, 25), dtype=np.uint64)
    lanes[:n_real] = inputs
    # bits[p, y, x, z]
    bits_np = ((lanes.reshape(n_perm, 5, 5, 1) >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8)
    A = torch.from_numpy(bits_np).to(device)
    rows_total = n_perm * KECCAK_ROUNDS
    dt = torch.int32
    trace = torch.zeros((K.WIDTH, n_perm, KECCAK_ROUNDS), dtype=dt, device=device)
    w16 = (1 << torch.arange(16, dtype=dt, device=device))

    def limbs(bits):   # (..., 64) -> (..., 4)
        return (bits.to(dt).reshape(*bits.shape[:-1], 4, 16) * w16).sum(-1)

    def put(col0, vals, r):   # vals: (n_perm, k) -> columns col0 .. col0 + k at round r
        trace[col0:col0 + vals.shape[1], :, r] = vals.t()

    pre_l = limbs(A).reshape(n_perm, 100)
    zsrc = torch.arange(64, device=device)
    for r in range(KECCAK_ROUNDS):
        trace[K.STEP_FLAGS + r, :, r] = 1
        put(K.PREIMAGE, pre_l, r)
        put(K.A, limbs(A).reshape(n_perm, 100), r)
        Cb = A[:, 0] ^ A[:, 1] ^ A[:, 2] ^ A[:, 3] ^ A[:, 4]                         # [p, x, z]
        Cp = Cb ^ Cb.roll(1, dims=1) ^ Cb.roll(-1, dims=1).roll(1, dims=2)            # C[x-1, z] and C[x+1, z-1]
        Ap = A ^ (Cb ^ Cp).unsqueeze(1)                                               # A' = A xor D, D = C xor C'
        put(K.C, Cb.reshape(n_perm, 320), r)
        put(K.C_PRIME, Cp.reshape(n_perm, 320), r)
        put(K.A_PRIME, Ap.reshape(n_perm, 1600), r)
        B = torch.empty_like(Ap)                                                      # B[p, y, x, z]
        for x in range(5):
            for y in range(5):
                sy, sx, _ = K.b_source(x, y, 0)
                a_, b_ = (x + 3 * y) % 5, x
                B[:, y, x] = Ap[:, sy, sx].roll(KECCAK_R[a_][b_], dims=1)
        App = B ^ ((1 - B.roll(-1, dims=2)) & B.roll(-2, dims=2))
        put(K.A_PRIME_PRIME, limbs(App).reshape(n_perm, 100), r)
        put(K.A_PP_00_BITS, App[:, 0, 0], r)
        rc = torch.tensor([(KECCAK_RC[r] >> z) & 1 for z in range(64)], dtype=torch.uint8, device=device)
        A = App.clone()
        A[:, 0, 0] = App[:, 0, 0] ^ rc
        put(K.A_PPP_00_LIMBS, limbs(A[:, 0, 0]), r)
    final_bits = A[:n_real].to(torch.int64).cpu().numpy().astype(np.uint64)
    final = [[int(sum(int(b) << z for z, b in enumerate(final_bits[p, i // 5, i % 5]))) for i in range(25)] for p in range(n_real)]
    src = trace.reshape(K.WIDTH, rows_total)
    flat = torch.empty((K.WIDTH, h), dtype=dt, device=device)
    for c0 in range(0, K.WIDTH, 256):   # (column blocks: the Montgomery conversion goes through int64)
        blk = src[c0:c0 + 256, :h]
        flat[c0:c0 + 256] = blk if mont_one is None else ((blk.to(torch.int64) * int(mont_one)) % P).to(dt)
    return flat.reshape(-1), final
