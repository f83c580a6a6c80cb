"""CPU tests: the oracle against the reference's golden vectors / in-tree constants and against
independent restatements (tests/py_ref.py). No GPU."""
import json
import os

import numpy as np
import pytest

import py_ref

HERE = os.path.dirname(os.path.abspath(__file__))
P = 2013265921


def golden(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


def test_field_constants_match_reference(orc, pkg):
    g = golden("field_constants.json")
    assert g["P"] == P == pkg.P
    # P * M == 1 mod 2^32 (fp.h:55-56)
    assert (g["P"] * g["MONTY_M"]) % (1 << 32) == 1
    assert g["R2"] == pow(2, 64, P)
    assert g["ONE_MONT"] == pow(2, 32, P) == int(orc.lib.orc_to_mont(1))
    gens = g["two_adic_generators"]
    for bits, w in enumerate(gens):
        assert int(orc.lib.orc_two_adic_generator(bits)) == w
        assert pow(w, 1 << bits, P) == 1
        if bits:
            assert pow(w, 1 << (bits - 1), P) == P - 1
            assert w * w % P == gens[bits - 1]
    # internal diagonal entries mean what poseidon2.cuh says they mean
    inv = lambda x: pow(x, P - 2, P)
    want = [-2, 1, 2, inv(2), 3, 4, -inv(2), -3, -4, inv(1 << 8), inv(4), inv(8), inv(1 << 27),
            -inv(1 << 8), -inv(16), -inv(1 << 27)]
    assert [x % P for x in want] == g["internal_diag16"]


def test_mont_roundtrip(orc, pkg):
    rng = np.random.default_rng(1)
    x = rng.integers(0, P, size=1000, dtype=np.uint64).astype(np.uint32)
    m = pkg.to_mont(x)
    assert all(int(orc.lib.orc_to_mont(int(a))) == int(b) for a, b in zip(x[:50], m[:50]))
    assert np.array_equal(pkg.from_mont(m), x)


def test_ext_field_axioms(orc):
    rng = np.random.default_rng(2)
    for _ in range(50):
        a = rng.integers(0, P, size=4, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, P, size=4, dtype=np.uint64).astype(np.uint32)
        ab = orc.ef_mul(a, b)
        assert np.array_equal(ab, orc.ef_mul(b, a))
        assert np.array_equal(orc.ef_mul(ab, orc.ef_inv(b)), a)
    # X^4 = 11 (fpext.h:28-37)
    x = np.array([0, 1, 0, 0], dtype=np.uint32)
    x2 = orc.ef_mul(x, x)
    assert list(orc.ef_mul(x2, x2)) == [11, 0, 0, 0]


def _p2_consts(orc):
    a = np.zeros(64, np.uint32); b = np.zeros(64, np.uint32)
    c = np.zeros(13, np.uint32); d = np.zeros(16, np.uint32)
    from oracle_lib import _p
    orc.lib.orc_poseidon2_constants(_p(a), _p(b), _p(c), _p(d))
    return [int(x) for x in a], [int(x) for x in b], [int(x) for x in c], [int(x) for x in d]


def test_poseidon2_matches_matrix_form(orc):
    ri, rt, rn, dg = _p2_consts(orc)
    assert dg == golden("field_constants.json")["internal_diag16"]
    # first external constant of p3-baby-bear's BABYBEAR_RC16_EXTERNAL_INITIAL (0x69cbb6af)
    assert ri[0] == 0x69CBB6AF
    rng = np.random.default_rng(3)
    for k in range(4):
        st = rng.integers(0, P, size=16, dtype=np.uint64).astype(np.uint32) if k else np.zeros(16, np.uint32)
        assert [int(x) for x in orc.permute(st)] == py_ref.permute(st, ri, rt, rn, dg)


def test_sponge_and_compress_semantics(orc):
    rng = np.random.default_rng(4)
    v = rng.integers(0, P, size=25, dtype=np.uint64).astype(np.uint32)
    # PaddingFreeSponge<16,8,8>: overwrite rate words, permute per chunk, stale tail kept
    st = np.zeros(16, np.uint32)
    for off in range(0, 25, 8):
        ch = v[off:off + 8]
        st[:len(ch)] = ch
        st = orc.permute(st)
    assert np.array_equal(orc.hash_slice(v), st[:8])
    assert np.array_equal(orc.hash_slice(np.zeros(0, np.uint32)), np.zeros(8, np.uint32))
    l, r = v[:8], v[8:16]
    assert np.array_equal(orc.compress(l, r), orc.permute(np.concatenate([l, r]))[:8])


def test_duplex_sponge_order(orc):
    """duplex_sponge.rs:62-83: 8th absorb permutes; samples come out from index 7 down."""
    t = orc.transcript()
    for i in range(8):
        t.observe(i + 1)
    st, a, s = t.state()
    assert (a, s) == (0, 8)
    assert np.array_equal(st, orc.permute(np.array(list(range(1, 9)) + [0] * 8, np.uint32)))
    assert t.sample() == int(st[7]) and t.sample() == int(st[6])
    t.observe(5)                      # pending absorb forces a permutation at next sample
    st2 = st.copy(); st2[0] = 5
    assert t.sample() == int(orc.permute(st2)[7])


def test_grind_is_minimum_witness(orc):
    t = orc.transcript()
    for i in range(3):
        t.observe(100 + i)
    bits = 6
    # brute force with forks of an identical transcript
    def ok(w):
        u = orc.transcript()
        for i in range(3):
            u.observe(100 + i)
        u.observe(w)
        return u.sample() & ((1 << bits) - 1) == 0
    w = t.grind(bits)
    assert ok(w) and not any(ok(x) for x in range(w))


@pytest.mark.parametrize("case", golden("stacking_literals.json")["cases"], ids=lambda c: c["name"])
def test_stacking_matches_reference_literals(orc, case):
    traces = [(np.array(c, np.uint32), 1, len(c)) for c in case["columns"]]
    out, width, layout = orc.stacked_matrix(case["l_skip"], case["n_stack"], traces)
    assert width == case["width"] and out.size == case["width"] * case["height"]
    assert list(out) == case["values"]
    if "mat_starts" in case:
        starts = [int(np.nonzero(layout[:, 0] == m)[0][0]) for m in range(len(traces))]
        assert starts == case["mat_starts"]


def test_dft_against_naive(orc):
    rng = np.random.default_rng(5)
    for lg in range(0, 6):
        n = 1 << lg
        a = rng.integers(0, P, size=n, dtype=np.uint64).astype(np.uint32)
        w = int(orc.lib.orc_two_adic_generator(lg))
        assert [int(x) for x in orc.dft(a)] == py_ref.naive_dft([int(x) for x in a], w)
        assert np.array_equal(orc.dft(orc.dft(a), inverse=True), a)


@pytest.mark.parametrize("l_skip,log_h,log_blowup", [(0, 3, 1), (2, 4, 1), (3, 3, 2), (2, 5, 0)])
def test_rs_code_against_naive(orc, l_skip, log_h, log_blowup):
    rng = np.random.default_rng(6)
    h, width = 1 << log_h, 3
    ev = rng.integers(0, P, size=width * h, dtype=np.uint64).astype(np.uint32)
    code = orc.rs_code_matrix(l_skip, log_blowup, ev, width, h)
    big = h << log_blowup
    w_big = int(orc.lib.orc_two_adic_generator(log_h + log_blowup))
    w_chunk = int(orc.lib.orc_two_adic_generator(l_skip))
    for j in range(width):
        msg = py_ref.eval_to_coeff(l_skip, [int(x) for x in ev[j * h:(j + 1) * h]], w_chunk)
        want = py_ref.naive_dft(msg + [0] * (big - h), w_big)
        assert [int(x) for x in code[j * big:(j + 1) * big]] == want


def test_merkle_tree_shape_and_paths(orc):
    """stacked_pcs.rs:413-485 (strided first layers) and :388-405, :516-540 (paths / opened rows);
    index arithmetic as pinned by crates/cpu-backend/src/merkle.rs:563-634."""
    rng = np.random.default_rng(7)
    width, height, rpq = 3, 32, 4
    m = rng.integers(0, P, size=width * height, dtype=np.uint64).astype(np.uint32)
    layers = orc.merkle_tree(m, width, height, rpq)
    qs = height // rpq
    assert [len(l) for l in layers] == [8, 4, 2, 1]
    rows = [orc.hash_slice(m.reshape(width, height)[:, r]) for r in range(height)]
    # digest_layers[0][y] is the root of the natural-order tree over rows y + t*qs
    for y in range(qs):
        leaves = [rows[y + t * qs] for t in range(rpq)]
        while len(leaves) > 1:
            leaves = [orc.compress(leaves[2 * i], leaves[2 * i + 1]) for i in range(len(leaves) // 2)]
        assert np.array_equal(layers[0][y], leaves[0])
    for l in range(1, len(layers)):
        for i in range(len(layers[l])):
            assert np.array_equal(layers[l][i], orc.compress(layers[l - 1][2 * i], layers[l - 1][2 * i + 1]))
    # a Merkle path recomputes the root
    for q in (0, 5, 7):
        cur, idx = layers[0][q], q
        for l in range(len(layers) - 1):
            sib = layers[l][idx ^ 1]
            cur = orc.compress(cur, sib) if idx % 2 == 0 else orc.compress(sib, cur)
            idx >>= 1
        assert np.array_equal(cur, layers[-1][0])
