// Host side of the prover: parameters, proving-key blob, stacked layout, proof byte writer,
// Fiat-Shamir transcript and the small polynomial algebra that stays on the CPU between kernel
// launches (the reference keeps exactly this on the host as well:
// crates/cpu-backend/src/logup_zerocheck/mod.rs:1468-1563).
//
// Mirrors (reference): crates/stark-backend/src/config.rs:51-65 (SystemParams),
// prover/stacked_pcs.rs:18-75,144-203 (StackedLayout), proof.rs:226-421 + codec.rs (encoding),
// transcript/duplex_sponge.rs + traits.rs (transcript), poly_common.rs / prover/poly.rs (algebra).
#pragma once
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "bb31.h"
#include "openvm_hip.h"
#include "poseidon2.h"

namespace ovm {
using bb::Fp;
using bb::FpExt;
using EF = bb::FpExt;

struct OvmFail : std::runtime_error {
    int code;
    OvmFail(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
#define OVM_THROW_IF(expr)                                                            \
    do {                                                                              \
        int _rc = (int)(expr);                                                        \
        if (_rc != 0) throw ovm::OvmFail(_rc, std::string(#expr) + " failed");        \
    } while (0)

inline unsigned log2_strict(size_t n) {
    unsigned l = 0;
    while ((size_t(1) << l) < n) l++;
    if ((size_t(1) << l) != n) throw OvmFail(1, "not a power of two");
    return l;
}
inline unsigned log2_ceil(size_t n) {
    unsigned l = 0;
    while ((size_t(1) << l) < n) l++;
    return l;
}
inline Fp fp(uint64_t x) { return Fp::from_canonical((uint32_t)(x % bb::P)); }
inline EF ef(Fp x) { return EF::from_base(x); }
inline EF ef_u64(uint64_t x) { return EF::from_base(fp(x)); }
inline bool is_zero(const EF &x) { return x == EF::zero(); }

// ---------------------------------------------------------------------------------------------
// Parameters and proving key
// ---------------------------------------------------------------------------------------------
struct Params {
    unsigned l_skip = 0, n_stack = 0, w_stack = 0, log_blowup = 0, k_whir = 0;
    unsigned mu_pow_bits = 0, query_phase_pow_bits = 0, folding_pow_bits = 0, logup_pow_bits = 0;
    unsigned max_constraint_degree = 0;
    std::vector<size_t> whir_num_queries;
};

enum NodeKind : uint32_t { N_VAR = 0, N_IS_FIRST, N_IS_LAST, N_IS_TRANSITION, N_CONST, N_ADD, N_SUB, N_NEG, N_MUL };
enum EntryKind : uint32_t { E_PREPROCESSED = 0, E_MAIN = 1, E_PUBLIC = 2, E_CHALLENGE = 3 };

struct Node {
    uint32_t kind = N_CONST, entry = 0, part_index = 0, offset = 0, index = 0, left = 0, right = 0;
    Fp constant = Fp::zero();
};
struct Interaction {
    std::vector<uint32_t> message;
    uint32_t count = 0, bus_index = 0, count_weight = 0;
};
struct AirKey {
    std::vector<Node> nodes;
    std::vector<uint32_t> constraint_idx;
    std::vector<Interaction> interactions;
    bool need_rot = false, is_required = true;
    unsigned max_constraint_degree = 0, num_public_values = 0;
    size_t common_main_width = 0;
    size_t preprocessed_width = 0;       // 0 = no preprocessed trace
    std::vector<size_t> cached_widths;   // cached main partitions (part_index 0..m-1; common main is part m)
    // host copy of the preprocessed trace from the blob's PREP section (column-major, Montgomery)
    std::vector<Fp> preprocessed_vals;
    size_t preprocessed_height = 0;
};
struct ProvingKey {
    Params params;
    p2::Digest vk_pre_hash;
    std::vector<AirKey> airs;
};

// Ranges the kernels and the index arithmetic are built for (two-adicity 27 bounds every domain).
inline void check_pcs_params(const Params &p) {
    if (p.l_skip > 10) throw OvmFail(1, "l_skip out of range (0..10)");
    if (p.l_skip + p.n_stack + p.log_blowup > 27) throw OvmFail(1, "l_skip + n_stack + log_blowup exceeds the two-adicity 27");
    if (p.k_whir == 0 || p.k_whir > 8 || p.k_whir > p.l_skip + p.n_stack) throw OvmFail(1, "k_whir out of range");
    if (p.log_blowup == 0 || p.log_blowup > 8) throw OvmFail(1, "log_blowup out of range (1..8)");
}

// Flat u32 proving-key blob, see include/openvm_hip.h. Field constants are canonical.
inline ProvingKey parse_pk_blob(const uint32_t *w, size_t n) {
    size_t i = 0;
    auto next = [&]() -> uint32_t {
        if (i >= n) throw OvmFail(1, "pk blob truncated");
        return w[i++];
    };
    if (next() != 0x4F564D31u) throw OvmFail(1, "bad pk blob magic");
    ProvingKey pk;
    Params &p = pk.params;
    p.l_skip = next();
    p.n_stack = next();
    p.w_stack = next();
    p.log_blowup = next();
    p.k_whir = next();
    p.mu_pow_bits = next();
    p.query_phase_pow_bits = next();
    p.folding_pow_bits = next();
    p.logup_pow_bits = next();
    p.max_constraint_degree = next();
    uint32_t rounds = next();
    if (rounds > 64) throw OvmFail(1, "pk blob: too many WHIR rounds");
    for (uint32_t r = 0; r < rounds; r++) p.whir_num_queries.push_back(next());
    check_pcs_params(p);
    if ((uint64_t)rounds * p.k_whir > p.l_skip + p.n_stack) throw OvmFail(1, "pk blob: WHIR rounds exceed the stacked height");
    if (p.mu_pow_bits > 30 || p.query_phase_pow_bits > 30 || p.folding_pow_bits > 30 || p.logup_pow_bits > 30)
        throw OvmFail(1, "pk blob: proof-of-work bits out of range");
    if (p.max_constraint_degree > 8) throw OvmFail(1, "pk blob: max_constraint_degree out of range");
    for (int k = 0; k < 8; k++) pk.vk_pre_hash.d[k] = fp(next());
    uint32_t num_airs = next();
    for (uint32_t a = 0; a < num_airs; a++) {
        AirKey air;
        air.need_rot = next() != 0;
        air.is_required = next() != 0;
        air.max_constraint_degree = next();
        air.num_public_values = next();
        uint32_t has_pre = next();
        air.preprocessed_width = next();
        if (!has_pre) air.preprocessed_width = 0;
        if (has_pre && air.preprocessed_width == 0) throw OvmFail(1, "preprocessed trace of width 0");
        uint32_t num_cached = next();
        for (uint32_t c = 0; c < num_cached; c++) air.cached_widths.push_back(next());
        air.common_main_width = next();
        uint32_t nn = next();
        for (uint32_t k = 0; k < nn; k++) {
            Node nd;
            nd.kind = next();
            nd.entry = next();
            nd.part_index = next();
            nd.offset = next();
            uint32_t x = next(), y = next();
            if (nd.kind == N_VAR) nd.index = x;
            else if (nd.kind == N_CONST) nd.constant = fp(x);
            else { nd.left = x; nd.right = y; }
            if (nd.kind > N_MUL) throw OvmFail(1, "bad node kind");
            if (nd.kind >= N_ADD && (nd.left >= k || (nd.kind != N_NEG && nd.right >= k)))
                throw OvmFail(1, "DAG not topologically ordered");
            air.nodes.push_back(nd);
        }
        uint32_t nc = next();
        for (uint32_t k = 0; k < nc; k++) air.constraint_idx.push_back(next());
        uint32_t ni = next();
        for (uint32_t k = 0; k < ni; k++) {
            Interaction it;
            it.bus_index = next();
            it.count_weight = next();
            it.count = next();
            uint32_t len = next();
            for (uint32_t j = 0; j < len; j++) it.message.push_back(next());
            air.interactions.push_back(it);
        }
        pk.airs.push_back(std::move(air));
    }
    // trailing section: 'PREP', then per AIR with a preprocessed trace: height, width*height values
    bool any_pre = false;
    for (auto &a : pk.airs) any_pre |= a.preprocessed_width != 0;
    if (any_pre) {
        if (next() != 0x50524550u) throw OvmFail(1, "pk blob: preprocessed section missing");
        for (auto &a : pk.airs) {
            if (!a.preprocessed_width) continue;
            a.preprocessed_height = next();
            if (a.preprocessed_height == 0 || (a.preprocessed_height & (a.preprocessed_height - 1)))
                throw OvmFail(1, "preprocessed trace height");
            if (a.preprocessed_height > (size_t(1) << (p.l_skip + p.n_stack)) ||
                a.preprocessed_width > (n - i) / a.preprocessed_height)  // (also rules out width * height overflowing)
                throw OvmFail(1, "pk blob: preprocessed trace larger than the blob");
            a.preprocessed_vals.resize(a.preprocessed_width * a.preprocessed_height);
            for (auto &v : a.preprocessed_vals) v = fp(next());
        }
    }
    return pk;
}

// ---------------------------------------------------------------------------------------------
// Stacked layout
// ---------------------------------------------------------------------------------------------
struct Slice {
    size_t col_idx = 0, row_idx = 0;
    unsigned log_height = 0;
    size_t len(unsigned l_skip) const { return size_t(1) << std::max(log_height, l_skip); }
    size_t stride(unsigned l_skip) const { return size_t(1) << (l_skip > log_height ? l_skip - log_height : 0); }
};
struct SortedCol {
    size_t mat_idx, col_idx;
    Slice slice;
};
struct Layout {
    unsigned l_skip = 0;
    size_t height = 0, width = 0;
    std::vector<SortedCol> cols;
    std::vector<size_t> mat_starts;
    Layout() {}
    Layout(unsigned l_skip_, unsigned log_stacked_height, const std::vector<std::pair<size_t, unsigned>> &sorted)
        : l_skip(l_skip_) {
        const size_t H = size_t(1) << log_stacked_height;
        size_t col = 0, row = 0;
        for (size_t m = 0; m < sorted.size(); m++) {
            mat_starts.push_back(cols.size());
            auto [w, lh] = sorted[m];
            if (w == 0) continue;
            if (lh > log_stacked_height) throw OvmFail(1, "LayoutHeightExceeded");
            size_t len = size_t(1) << std::max(lh, l_skip);
            for (size_t j = 0; j < w; j++) {
                if (row + len > H) {
                    if (row != H) throw OvmFail(1, "LayoutRowOverflow");
                    col++;
                    row = 0;
                }
                cols.push_back({m, j, Slice{col, row, lh}});
                row += len;
            }
        }
        height = H;
        width = col + (row != 0 ? 1 : 0);
    }
};

// ---------------------------------------------------------------------------------------------
// Proof bytes
// ---------------------------------------------------------------------------------------------
struct ByteWriter {
    std::vector<uint8_t> buf;
    void u8(uint8_t x) { buf.push_back(x); }
    void u32(uint32_t x) {
        for (int i = 0; i < 4; i++) buf.push_back((uint8_t)(x >> (8 * i)));
    }
    void usize(size_t x) { u32((uint32_t)x); }
    void f(Fp x) { u32(x.canonical()); }
    void e(const EF &x) {
        for (int i = 0; i < 4; i++) f(x.c[i]);
    }
    void digest(const p2::Digest &d) {
        for (int i = 0; i < 8; i++) f(d.d[i]);
    }
    void e_slice(const std::vector<EF> &v) {
        usize(v.size());
        for (auto &x : v) e(x);
    }
};

// ---------------------------------------------------------------------------------------------
// Transcript: host duplex sponge + device proof-of-work
// ---------------------------------------------------------------------------------------------
struct Transcript {
    p2::Sponge s;
    hipStream_t stream = nullptr;
    Transcript() { s.init(); }
    void observe(Fp x) { s.observe(x); }
    void observe_ext(const EF &x) { s.observe_ext(x); }
    void observe_digest(const p2::Digest &d) { s.observe_digest(d); }
    Fp sample() { return s.sample(); }
    EF sample_ext() { return s.sample_ext(); }
    uint32_t sample_bits(unsigned bits) { return s.sample_bits(bits); }
    // minimum witness (device search), then absorbed like check_witness does
    Fp grind(unsigned bits) {
        if (bits == 0) return Fp::zero();
        if (bits <= 6) {
            // ~2^bits candidates: a host search (about 0.25 us per candidate) beats a device launch plus
            // two copies and a synchronisation (~85 us); same minimum witness
            const uint32_t w = p2::host_grind_min_witness(s, bits);  // (eight candidates per step: host_poseidon2.cpp)
            if (w == ~0u) throw OvmFail(1, "grind: no witness");
            s.observe(Fp::from_canonical(w));
            if (s.sample_bits(bits) != 0) throw OvmFail(1, "grind: witness does not verify");
            return Fp::from_canonical(w);
        }
        ovm_sponge_t st;
        for (int i = 0; i < 16; i++) st.state[i] = s.state[i].v;
        st.absorb_idx = s.absorb_idx;
        st.sample_idx = s.sample_idx;
        uint32_t w = 0;
        OVM_THROW_IF(ovm_hip_sponge_grind(&st, bits, &w, (ovm_stream_t)stream));
        Fp wf = Fp::from_canonical(w);
        s.observe(wf);
        if (s.sample_bits(bits) != 0) throw OvmFail(1, "grind: witness does not verify");
        return wf;
    }
};

// ---------------------------------------------------------------------------------------------
// Small host algebra (sizes <= a few hundred)
// ---------------------------------------------------------------------------------------------
template <class T>
inline std::vector<T> powers_of(T base, size_t n) {
    std::vector<T> v(n);
    T cur = T::one();
    for (size_t i = 0; i < n; i++) {
        v[i] = cur;
        cur = cur * base;
    }
    return v;
}

// out[k] = sum_i a[i] w^(ik), naive (n <= 256)
inline std::vector<EF> small_dft(const std::vector<EF> &a, bool inverse) {
    size_t n = a.size();
    unsigned lg = log2_strict(n);
    Fp w = bb::two_adic_generator(lg);
    if (inverse) w = w.inv();
    std::vector<Fp> wp = powers_of(w, n);
    std::vector<EF> out(n, EF::zero());
    for (size_t k = 0; k < n; k++) {
        EF acc = EF::zero();
        for (size_t i = 0; i < n; i++) acc += a[i] * wp[(i * k) % n];
        out[k] = acc;
    }
    if (inverse) {
        Fp ninv = fp(n).inv();
        for (auto &x : out) x = x * ninv;
    }
    return out;
}

template <class C>
inline EF horner(const std::vector<C> &coeffs, EF x) {
    EF acc = EF::zero();
    for (size_t i = coeffs.size(); i-- > 0;) acc = acc * x + coeffs[i];
    return acc;
}
inline EF horner_fp(const std::vector<EF> &coeffs, Fp x) {
    EF acc = EF::zero();
    for (size_t i = coeffs.size(); i-- > 0;) acc = acc * x + coeffs[i];
    return acc;
}

inline EF eq1(EF x, EF y) { return EF::one() - y - x + (x * y).dbl(); }
inline EF eq1b(EF x, bool b) { return b ? x : EF::one() - x; }

template <class T>
inline T eq_uni(unsigned l, T x, T y) {
    T res = T::one(), xp = x, yp = y;
    for (unsigned i = 0; i < l; i++) {
        res = (xp + yp) * res + (xp - T::one()) * (yp - T::one());
        xp = xp.square();
        yp = yp.square();
    }
    return res * Fp::one().halve().pow(l);
}
template <class T>
inline T eq_uni_at_one(unsigned l, T x) {
    T res = T::one(), xp = x;
    for (unsigned i = 0; i < l; i++) {
        res = res * (xp + T::one());
        xp = xp.square();
    }
    return res * Fp::one().halve().pow(l);
}
template <class T>
inline T in_uni(unsigned l_skip, int n, T z) {
    if (n < 0) return eq_uni_at_one((unsigned)(-n), z.exp_power_of_2(l_skip + n));
    return T::one();
}
inline std::vector<EF> eq_hypercube(const std::vector<EF> &x) {
    std::vector<EF> out(size_t(1) << x.size());
    out[0] = EF::one();
    for (size_t i = 0; i < x.size(); i++)
        for (size_t y = 0; y < (size_t(1) << i); y++) {
            EF lo = out[y];
            out[y + (size_t(1) << i)] = lo * x[i];
            out[y] = lo * (EF::one() - x[i]);
        }
    return out;
}

// coefficients of the interpolant through (points[i], evals[i]), points in the base field
inline std::vector<EF> lagrange_coeffs(const std::vector<Fp> &points, const std::vector<EF> &evals) {
    size_t n = points.size();
    std::vector<EF> coeffs(n, EF::zero());
    // master polynomial M(x) = prod (x - p_j)
    std::vector<Fp> m{Fp::one()};
    for (size_t j = 0; j < n; j++) {
        m.push_back(Fp::zero());
        for (size_t k = m.size() - 1; k >= 1; k--) m[k] = m[k - 1] - points[j] * m[k];
        m[0] = -(points[j] * m[0]);
    }
    for (size_t i = 0; i < n; i++) {
        // q = M / (x - p_i) by synthetic division; denominator = q(p_i)
        std::vector<Fp> q(n);
        q[n - 1] = m[n];
        for (size_t k = n - 1; k-- > 0;) q[k] = m[k + 1] + points[i] * q[k + 1];
        Fp den = Fp::zero();
        for (size_t k = n; k-- > 0;) den = den * points[i] + q[k];
        EF scale = evals[i] * den.inv();
        for (size_t k = 0; k < n; k++) coeffs[k] += scale * q[k];
    }
    return coeffs;
}

// Interpolate the polynomial of degree < N*d from its values on the d cosets g^(k+1) * D, |D| = N:
// evals[k][z] = f(g^(k+1) * omega^z)  (reference: prover/poly.rs:621-683).
inline std::vector<EF> from_coset_evals(const std::vector<std::vector<EF>> &evals, unsigned l) {
    size_t d = evals.size(), N = size_t(1) << l;
    if (d == 0) return {};
    Fp g = fp(bb::GENERATOR);
    std::vector<std::vector<EF>> rem(d);
    std::vector<Fp> ys(d);
    for (size_t k = 0; k < d; k++) {
        Fp shift = g.pow(k + 1);
        std::vector<EF> c = small_dft(evals[k], true);
        Fp sinv = shift.inv(), sp = Fp::one();
        for (size_t i = 0; i < N; i++) {
            c[i] = c[i] * sp;
            sp *= sinv;
        }
        rem[k] = std::move(c);
        ys[k] = shift.exp_power_of_2(l);
    }
    std::vector<EF> out(N * d);
    for (size_t t = 0; t < N; t++) {
        std::vector<EF> vals(d);
        for (size_t k = 0; k < d; k++) vals[k] = rem[k][t];
        std::vector<EF> pc = lagrange_coeffs(ys, vals);
        for (size_t k = 0; k < d; k++) out[k * N + t] = pc[k];
    }
    return out;
}

}  // namespace ovm
