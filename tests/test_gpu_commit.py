"""GPU parity tests (through the C ABI) for the commit path: Poseidon2, eval->coeff, NTT, RS
encode, Merkle tree, grind -- bit-exact against the CPU oracle on the same seeded inputs."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2013265921


def rand_fp(rng, n):
    return rng.integers(0, P, size=n, dtype=np.uint64).astype(np.uint32)


def brev(i, bits):
    r = 0
    for b in range(bits):
        r |= ((i >> b) & 1) << (bits - 1 - b)
    return r


def test_poseidon2_permute(dev, pkg, orc):
    """The device permutation (poseidon2.h: lazy::permute -- linear layers summed unreduced in 64-bit registers, one signed
    reduction per word, the state's scale tracked at compile time) against the oracle's plain canonical one on EVERY state
    of: the range extremes the lazily reduced sums have to hold (all zero, all P - 1, one extreme word among zeros, one zero
    among extremes, alternating patterns, words around P / 2) and 900 random states."""
    rng = np.random.default_rng(10)
    n = 1000
    st = rand_fp(rng, 16 * n).reshape(n, 16)
    hi = P - 1
    st[0] = 0
    st[1] = hi
    for i in range(16):
        st[2 + i] = 0
        st[2 + i][i] = hi
        st[18 + i] = hi
        st[18 + i][i] = 0
    for k in range(34, 100):
        pat = [(hi if (k >> (i & 3)) & 1 else (P // 2 + i if k & 32 else 0)) for i in range(16)]
        st[k] = np.array(pat, dtype=np.uint32)
    d = dev.upload(pkg.to_mont(st.reshape(-1)))
    dev.poseidon2_permute(d, n)
    got = pkg.from_mont(dev.download(d)).reshape(n, 16)
    for i in range(n):
        assert np.array_equal(got[i], orc.permute(st[i])), i


@pytest.mark.parametrize("l_skip,log_h", [(0, 5), (1, 3), (2, 2), (4, 10), (4, 15), (6, 14)])
def test_eval_to_coeff_and_rs_encode(dev, pkg, orc, l_skip, log_h):
    rng = np.random.default_rng(11)
    width, h = 3, 1 << log_h
    ev = rand_fp(rng, width * h)
    for log_blowup in (1, 2):
        d_ev = dev.upload(pkg.to_mont(ev))
        d_code = dev.rs_encode(d_ev, width, log_h, l_skip, log_blowup)
        got = pkg.from_mont(dev.download(d_code)).reshape(width, h << log_blowup)
        want = orc.rs_code_matrix(l_skip, log_blowup, ev, width, h).reshape(width, h << log_blowup)
        L = log_h + log_blowup
        perm = pkg.bitrev(np.arange(1 << L), L).astype(np.int64)
        # physical row p holds natural index bitrev(p)
        assert np.array_equal(got, want[:, perm])


@pytest.mark.parametrize("log_n", [0, 1, 7, 13, 14, 20])
def test_ntt_dif_base(dev, pkg, orc, log_n):
    rng = np.random.default_rng(12)
    n = 1 << log_n
    a = rand_fp(rng, 2 * n)
    d = dev.upload(pkg.to_mont(a))
    out = dev.empty(2 * n)
    dev.ntt_dif(d, out, 2, log_n)
    got = pkg.from_mont(dev.download(out)).reshape(2, n)
    perm = pkg.bitrev(np.arange(n), log_n).astype(np.int64)
    for j in range(2):
        assert np.array_equal(got[j], orc.dft(a[j * n:(j + 1) * n])[perm])


def test_ntt_dif_ext_is_coefficientwise(dev, pkg, orc):
    rng = np.random.default_rng(13)
    log_n, n = 12, 1 << 12
    a = rand_fp(rng, 4 * n)          # n extension elements, [u32;4] each
    d = dev.upload(pkg.to_mont(a))
    out = dev.empty(8 * n)           # zero-extended x2
    dev.ntt_dif(d, out, 1, log_n + 1, log_n_in=log_n, ext=True)
    got = pkg.from_mont(dev.download(out)).reshape(2 * n, 4)
    perm = pkg.bitrev(np.arange(2 * n), log_n + 1).astype(np.int64)
    for k in range(4):
        col = np.concatenate([a.reshape(n, 4)[:, k], np.zeros(n, np.uint32)])
        assert np.array_equal(got[:, k], orc.dft(col)[perm])


@pytest.mark.parametrize("width,log_rows,k", [(1, 4, 0), (3, 6, 2), (25, 10, 4), (8, 9, 4), (17, 8, 3)])
def test_merkle_tree_base(dev, pkg, orc, width, log_rows, k):
    rng = np.random.default_rng(14)
    rows = 1 << log_rows
    nat = rand_fp(rng, width * rows).reshape(width, rows)           # natural row order
    perm = pkg.bitrev(np.arange(rows), log_rows).astype(np.int64)
    phys = nat[:, perm]                                              # physical = bit-reversed
    d = dev.upload(pkg.to_mont(phys.reshape(-1)))
    layers = pkg.from_mont(dev.download(dev.merkle_tree(d, width, log_rows, k))).reshape(-1, 8)
    want = orc.merkle_tree(nat.reshape(-1), width, rows, 1 << k)
    off = 0
    for j, wl in enumerate(want):
        n = len(wl)
        lg = log_rows - k - j
        idx = pkg.bitrev(np.arange(n), lg).astype(np.int64)
        assert np.array_equal(layers[off:off + n][idx], wl), f"layer {j}"
        off += n


def test_merkle_tree_ext(dev, pkg, orc):
    rng = np.random.default_rng(15)
    log_rows, k, rows = 9, 4, 1 << 9
    nat = rand_fp(rng, rows * 4).reshape(rows, 4)
    perm = pkg.bitrev(np.arange(rows), log_rows).astype(np.int64)
    d = dev.upload(pkg.to_mont(nat[perm].reshape(-1)))
    layers = pkg.from_mont(dev.download(dev.merkle_tree(d, 1, log_rows, k, ext=True))).reshape(-1, 8)
    want = orc.merkle_tree(nat.reshape(-1), 1, rows, 1 << k, ext=True)
    assert np.array_equal(layers[-1], want[-1][0])
    idx = pkg.bitrev(np.arange(len(want[0])), log_rows - k).astype(np.int64)
    assert np.array_equal(layers[:len(want[0])][idx], want[0])


@pytest.mark.parametrize("bits,n_obs", [(1, 0), (8, 3), (12, 7), (16, 5), (20, 2)])
def test_grind_minimum_witness(dev, pkg, orc, bits, n_obs):
    t = orc.transcript()
    for i in range(n_obs):
        t.observe(12345 + i)
    st, a, s = t.state()
    sp = pkg.SpongeState()
    m = pkg.to_mont(st)
    for i in range(16):
        sp.state[i] = int(m[i])
    sp.absorb_idx, sp.sample_idx = a, s
    assert dev.grind(sp, bits) == t.grind(bits)


def test_commit_full_size_properties(dev, pkg):
    """BASELINE config D shape (2^24 x 25 stacked, blowup 2): size-independent properties --
    linearity of the encoder and the codeword's restriction to the message domain."""
    rng = np.random.default_rng(16)
    log_h, l_skip, width = 20, 4, 4      # 2^20 x 4 keeps the check under a second of GPU time
    h = 1 << log_h
    a, b = rand_fp(rng, width * h), rand_fp(rng, width * h)
    s = ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    ca = pkg.from_mont(dev.download(dev.rs_encode(dev.upload(pkg.to_mont(a)), width, log_h, l_skip, 1)))
    cb = pkg.from_mont(dev.download(dev.rs_encode(dev.upload(pkg.to_mont(b)), width, log_h, l_skip, 1)))
    cs = pkg.from_mont(dev.download(dev.rs_encode(dev.upload(pkg.to_mont(s)), width, log_h, l_skip, 1)))
    assert np.array_equal(cs, ((ca.astype(np.uint64) + cb) % P).astype(np.uint32))


def test_p3_known_answer_through_the_hip_layers(dev, pkg, orc):
    """Plonky3's published Poseidon2 vector (tests/golden/poseidon2_p3_kat.json: RNG-drawn round constants) through the
    PRODUCT's S-box and linear-layer code on the GPU (ovm_hip_poseidon2_permute_rc), and random states against the oracle
    under the same constants; the default permutation is the same layer code over the in-tree default tables."""
    import json
    import os
    import p3_rng
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poseidon2_p3_kat.json")))
    ri, rt, rn = p3_rng.poseidon2_constants_from_rng_128(kat["seed"])
    d_rc = dev.upload(pkg.to_mont(np.array(ri + rt + rn, dtype=np.uint32)))
    rng = np.random.default_rng(5)
    states = rng.integers(0, P, size=(257, 16), dtype=np.uint64).astype(np.uint32)
    states[0] = kat["input"]
    d = dev.upload(pkg.to_mont(states.reshape(-1)))
    dev.poseidon2_permute_rc(d, 257, d_rc)
    got = pkg.from_mont(dev.download(d)).reshape(257, 16)
    assert [int(x) for x in got[0]] == kat["output"]
    for i in (1, 2, 100, 256):
        assert np.array_equal(got[i], orc.permute_rc(states[i], ri, rt, rn))
