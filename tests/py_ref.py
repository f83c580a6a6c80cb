"""Independent big-int restatements used to cross-check the C++ oracle (test infrastructure).

Poseidon2 here is written from the layer DEFINITIONS as explicit 16x16 matrices
(M_E = circ(2*M4, M4, M4, M4), M_I = J + diag(V)) rather than the additive shortcuts both the
oracle and the product use, so agreement checks the linear-layer code, not just the constants.
Structure follows crates/cuda-common/include/poseidon2.cuh:86-202 of the reference.
"""
P = 2013265921

M4 = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]


def m_ext():
    m = [[0] * 16 for _ in range(16)]
    for bi in range(4):
        for bj in range(4):
            k = 2 if bi == bj else 1
            for i in range(4):
                for j in range(4):
                    m[4 * bi + i][4 * bj + j] = k * M4[i][j] % P
    return m


def m_int(diag):
    return [[(1 + (diag[i] if i == j else 0)) % P if i == j else 1 for j in range(16)]
            for i in range(16)]


def matvec(m, v):
    return [sum(m[i][j] * v[j] for j in range(16)) % P for i in range(16)]


def permute(state, rc_init, rc_term, rc_int, diag):
    me, mi = m_ext(), m_int(diag)
    s = matvec(me, [int(x) for x in state])
    for r in range(4):
        s = [pow((s[i] + rc_init[16 * r + i]) % P, 7, P) for i in range(16)]
        s = matvec(me, s)
    for r in range(13):
        s[0] = pow((s[0] + rc_int[r]) % P, 7, P)
        s = matvec(mi, s)
    for r in range(4):
        s = [pow((s[i] + rc_term[16 * r + i]) % P, 7, P) for i in range(16)]
        s = matvec(me, s)
    return s


def naive_dft(a, w):
    n = len(a)
    return [sum(a[i] * pow(w, i * k, P) for i in range(n)) % P for k in range(n)]


def eval_to_coeff(l_skip, evals, w_chunk):
    """poly.rs:325-348 restated naively: per chunk inverse DFT then subset-zeta."""
    c = 1 << l_skip
    out = []
    winv = pow(w_chunk, P - 2, P)
    ninv = pow(c, P - 2, P)
    for off in range(0, len(evals), c):
        ch = [x * ninv % P for x in naive_dft(evals[off:off + c], winv)]
        for b in range(l_skip):
            for i in range(c):
                if i >> b & 1:
                    ch[i] = (ch[i] + ch[i ^ (1 << b)]) % P
        out += ch
    return out
