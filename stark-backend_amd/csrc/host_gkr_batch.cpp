// Host side of the LogUp-GKR layer sumchecks, vectorised: the records a rank hands to the host (the last <= 64 of every
// layer's table, the pairs that straddle ranks) are summed and folded here between two device round trips -- ~950
// extension-field products per layer tail, 43 us of one core per tail in scalar code (26 tails per config D proof,
// idle_gaps_r03.txt: "gkr_edge_kernel 1.13 ms over 26 gaps"). Eight pairs at a time in AVX2: limb-planar 8-lane values,
// lazy 64-bit accumulation of the 16 limb products of an extension-field multiply (one reduction per output limb, as
// bb::FpExt::operator* does in scalar code), selected at run time like the transcript's permutation
// (host_poseidon2.cpp). Exact field arithmetic: the results are the scalar ones, bit for bit
// (ovm_hip_host_gkr_batch_selftest).
//
// Reference: the sums are those of crates/stark-backend/src/prover/logup_zerocheck/fractional_sumcheck_gkr.rs (the
// per-pair terms of a layer's sumcheck round); who computes which pair where is this backend's own (gkr_shard.h).
#if !defined(__HIP_DEVICE_COMPILE__)
#include "gkr_shard.h"

#include <cstdlib>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace ovm {

static void pair_sums_scalar(const GkrRec *A, const GkrRec *B, const EF *e, size_t n, const EF &lambda, unsigned n_points, EF *S) {
    for (size_t i = 0; i < n; i++) rec_pair_sums(A[i], B[i], e[i], lambda, n_points, S);
}
static void fold_scalar(const GkrRec *A, const GkrRec *B, size_t n, const EF &r, GkrRec *out) {
    for (size_t i = 0; i < n; i++) out[i] = rec_fold(A[i], B[i], r);
}
static void eq_scalar(const EF *pt, unsigned nv, const uint64_t *y, size_t n, EF *out) {
    for (size_t i = 0; i < n; i++) out[i] = eq_at(pt, nv, y[i]);
}

#if defined(__x86_64__)
#define OVM_AVX2 __attribute__((target("avx2"))) inline
namespace {
struct EF8 {
    __m256i c[4];  // limb k of eight extension-field values
};
OVM_AVX2 __m256i p8() { return _mm256_set1_epi32((int)bb::P); }
OVM_AVX2 __m256i add_p(__m256i a, __m256i b) {
    const __m256i t = _mm256_add_epi32(a, b);
    return _mm256_min_epu32(t, _mm256_sub_epi32(t, p8()));
}
OVM_AVX2 __m256i sub_p(__m256i a, __m256i b) {
    const __m256i t = _mm256_sub_epi32(a, b);
    return _mm256_min_epu32(t, _mm256_add_epi32(t, p8()));
}
OVM_AVX2 EF8 add(const EF8 &a, const EF8 &b) {
    EF8 r;
    for (int k = 0; k < 4; k++) r.c[k] = add_p(a.c[k], b.c[k]);
    return r;
}
OVM_AVX2 EF8 sub(const EF8 &a, const EF8 &b) {
    EF8 r;
    for (int k = 0; k < 4; k++) r.c[k] = sub_p(a.c[k], b.c[k]);
    return r;
}
// 8 Montgomery products (as host_poseidon2.cpp)
OVM_AVX2 __m256i mont_mul(__m256i a, __m256i b) {
    const __m256i P8 = p8(), M8 = _mm256_set1_epi32((int)bb::MONTY_M);
    const __m256i a_odd = _mm256_srli_epi64(a, 32), b_odd = _mm256_srli_epi64(b, 32);
    const __m256i pe = _mm256_mul_epu32(a, b), po = _mm256_mul_epu32(a_odd, b_odd);
    const __m256i ue = _mm256_mul_epu32(_mm256_mul_epu32(pe, M8), P8), uo = _mm256_mul_epu32(_mm256_mul_epu32(po, M8), P8);
    const __m256i de = _mm256_sub_epi64(pe, ue), dodd = _mm256_sub_epi64(po, uo);
    const __m256i r = _mm256_blend_epi32(_mm256_srli_epi64(de, 32), dodd, 0xAA);
    return _mm256_min_epu32(r, _mm256_add_epi32(r, P8));
}
// x < 4 P^2 in each 64-bit lane  ->  x * 2^-32 mod P in the lane's low word (bb::mont_reduce_wide)
OVM_AVX2 __m256i reduce_wide64(__m256i x) {
    const __m256i P8 = p8(), M8 = _mm256_set1_epi32((int)bb::MONTY_M);
    const __m256i u = _mm256_mul_epu32(_mm256_mul_epu32(x, M8), P8);  // t * P, t = lo(x) * M mod 2^32
    __m256i xh = _mm256_srli_epi64(x, 32);
    const __m256i uh = _mm256_srli_epi64(u, 32);
    xh = _mm256_min_epu32(xh, _mm256_sub_epi32(xh, P8));  // hi(x) may reach ~1.9 P
    const __m256i r = _mm256_sub_epi32(xh, uh);
    return _mm256_min_epu32(r, _mm256_add_epi32(r, P8));  // (the upper word of each lane is junk: masked by the caller)
}
OVM_AVX2 __m256i dot4(__m256i x0, __m256i y0, __m256i x1, __m256i y1, __m256i x2, __m256i y2, __m256i x3, __m256i y3) {
    return _mm256_add_epi64(_mm256_add_epi64(_mm256_mul_epu32(x0, y0), _mm256_mul_epu32(x1, y1)),
                            _mm256_add_epi64(_mm256_mul_epu32(x2, y2), _mm256_mul_epu32(x3, y3)));
}
OVM_AVX2 __m256i pack(__m256i even, __m256i odd) {
    const __m256i lo_mask = _mm256_set1_epi64x(0xffffffffll);
    return _mm256_or_si256(_mm256_and_si256(reduce_wide64(even), lo_mask), _mm256_slli_epi64(reduce_wide64(odd), 32));
}
// a * b for eight pairs: per output limb ONE 64-bit sum of four raw products (even and odd lanes apart) and one reduction
OVM_AVX2 EF8 mul(const EF8 &a, const EF8 &b) {
    const __m256i W = _mm256_set1_epi32((int)bb::cx_to_mont(11));
    __m256i be[4], bo[4], we[4], wo[4], ae[4], ao[4];
    for (int k = 0; k < 4; k++) {
        ae[k] = a.c[k], ao[k] = _mm256_srli_epi64(a.c[k], 32);
        be[k] = b.c[k], bo[k] = _mm256_srli_epi64(b.c[k], 32);
    }
    for (int k = 1; k < 4; k++) {
        const __m256i w = mont_mul(b.c[k], W);  // 11 * b_k
        we[k] = w, wo[k] = _mm256_srli_epi64(w, 32);
    }
    EF8 r;
    // c0 = a0 b0 + a1 w3 + a2 w2 + a3 w1 ; c1 = a0 b1 + a1 b0 + a2 w3 + a3 w2 ; c2 = a0 b2 + a1 b1 + a2 b0 + a3 w3 ;
    // c3 = a0 b3 + a1 b2 + a2 b1 + a3 b0        (w_k = 11 b_k: X^4 = 11)
    r.c[0] = pack(dot4(ae[0], be[0], ae[1], we[3], ae[2], we[2], ae[3], we[1]), dot4(ao[0], bo[0], ao[1], wo[3], ao[2], wo[2], ao[3], wo[1]));
    r.c[1] = pack(dot4(ae[0], be[1], ae[1], be[0], ae[2], we[3], ae[3], we[2]), dot4(ao[0], bo[1], ao[1], bo[0], ao[2], wo[3], ao[3], wo[2]));
    r.c[2] = pack(dot4(ae[0], be[2], ae[1], be[1], ae[2], be[0], ae[3], we[3]), dot4(ao[0], bo[2], ao[1], bo[1], ao[2], bo[0], ao[3], wo[3]));
    r.c[3] = pack(dot4(ae[0], be[3], ae[1], be[2], ae[2], be[1], ae[3], be[0]), dot4(ao[0], bo[3], ao[1], bo[2], ao[2], bo[1], ao[3], bo[0]));
    return r;
}
OVM_AVX2 EF8 splat(const EF &x) {
    EF8 r;
    for (int k = 0; k < 4; k++) r.c[k] = _mm256_set1_epi32((int)x.c[k].v);
    return r;
}
// eight values taken `stride` bytes apart (lanes past n are zero)
OVM_AVX2 EF8 load8(const EF *first, size_t stride_bytes, size_t n) {
    alignas(32) uint32_t t[4][8] = {};
    for (size_t i = 0; i < n && i < 8; i++) {
        const EF *p = reinterpret_cast<const EF *>(reinterpret_cast<const char *>(first) + i * stride_bytes);
        for (int k = 0; k < 4; k++) t[k][i] = p->c[k].v;
    }
    EF8 r;
    for (int k = 0; k < 4; k++) r.c[k] = _mm256_load_si256(reinterpret_cast<const __m256i *>(t[k]));
    return r;
}
OVM_AVX2 void store8(const EF8 &v, EF *first, size_t stride_bytes, size_t n) {
    alignas(32) uint32_t t[4][8];
    for (int k = 0; k < 4; k++) _mm256_store_si256(reinterpret_cast<__m256i *>(t[k]), v.c[k]);
    for (size_t i = 0; i < n && i < 8; i++) {
        EF *p = reinterpret_cast<EF *>(reinterpret_cast<char *>(first) + i * stride_bytes);
        for (int k = 0; k < 4; k++) p->c[k] = Fp::raw(t[k][i]);
    }
}
// the sum of the (first n) lanes, as one value
OVM_AVX2 EF hsum(const EF8 &v, size_t n) {
    alignas(32) uint32_t t[4][8];
    for (int k = 0; k < 4; k++) _mm256_store_si256(reinterpret_cast<__m256i *>(t[k]), v.c[k]);
    EF r = EF::zero();
    for (size_t i = 0; i < n && i < 8; i++) {
        EF x;
        for (int k = 0; k < 4; k++) x.c[k] = Fp::raw(t[k][i]);
        r = r + x;
    }
    return r;
}

__attribute__((target("avx2"))) void pair_sums_avx2(const GkrRec *A, const GkrRec *B, const EF *e, size_t n, const EF &lambda,
                                                    unsigned n_points, EF *S) {
    const EF8 lam = splat(lambda);
    for (size_t i0 = 0; i0 < n; i0 += 8) {
        const size_t m = n - i0 < 8 ? n - i0 : 8;
        EF8 v[4], d[4];
        for (int k = 0; k < 4; k++) {
            const EF8 a = load8(&A[i0].v[k], sizeof(GkrRec), m);
            v[k] = load8(&B[i0].v[k], sizeof(GkrRec), m);  // X = 1
            d[k] = sub(v[k], a);
        }
        const EF8 w = load8(e + i0, sizeof(EF), m);  // (lanes past m: weight zero)
        for (unsigned x = 1; x <= n_points; x++) {
            // (p0 q1 + q0 (p1 + lambda q1)) * e
            const EF8 term = mul(add(mul(v[0], v[3]), mul(v[1], add(v[2], mul(lam, v[3])))), w);
            S[x - 1] += hsum(term, m);
            for (int k = 0; k < 4; k++) v[k] = add(v[k], d[k]);
        }
    }
}
__attribute__((target("avx2"))) void fold_avx2(const GkrRec *A, const GkrRec *B, size_t n, const EF &r, GkrRec *out) {
    const EF8 rr = splat(r);
    for (size_t i0 = 0; i0 < n; i0 += 8) {
        const size_t m = n - i0 < 8 ? n - i0 : 8;
        for (int k = 0; k < 4; k++) {
            const EF8 a = load8(&A[i0].v[k], sizeof(GkrRec), m), b = load8(&B[i0].v[k], sizeof(GkrRec), m);
            store8(add(a, mul(sub(b, a), rr)), &out[i0].v[k], sizeof(GkrRec), m);
        }
    }
}
// eq(pt, y) = prod_k (y_k ? pt_k : 1 - pt_k): the factor of variable k is one of two splatted values, picked per lane
__attribute__((target("avx2"))) void eq_avx2(const EF *pt, unsigned nv, const uint64_t *y, size_t n, EF *out) {
    for (size_t i0 = 0; i0 < n; i0 += 8) {
        const size_t m = n - i0 < 8 ? n - i0 : 8;
        EF8 acc = splat(EF::one());
        for (unsigned k = 0; k < nv; k++) {
            const EF8 one = splat(pt[k]), zero = splat(EF::one() - pt[k]);
            alignas(32) uint32_t sel[8] = {};
            for (size_t i = 0; i < m; i++) sel[i] = ((y[i0 + i] >> k) & 1) ? 0xffffffffu : 0u;
            const __m256i s = _mm256_load_si256(reinterpret_cast<const __m256i *>(sel));
            EF8 f;
            for (int c = 0; c < 4; c++) f.c[c] = _mm256_blendv_epi8(zero.c[c], one.c[c], s);
            acc = mul(acc, f);
        }
        store8(acc, out + i0, sizeof(EF), m);
    }
}
bool have_avx2() {
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}
}  // namespace
#endif  // __x86_64__

static bool g_force_scalar = false;  // (self-test)
static bool scalar_only() {
#if defined(OVM_TEST_HOOKS)
    static const bool off = [] {
        const char *e = getenv("OVM_HOST_GKR_SIMD");  // =0: scalar host arithmetic (A/B)
        return e && e[0] == '0';
    }();
    return off || g_force_scalar;
#else
    return g_force_scalar;
#endif
}

void gkr_pair_sums_batch(const GkrRec *A, const GkrRec *B, const EF *e, size_t n, const EF &lambda, unsigned n_points, EF *S) {
#if defined(__x86_64__)
    if (n >= 4 && !scalar_only() && have_avx2()) return pair_sums_avx2(A, B, e, n, lambda, n_points, S);
#endif
    pair_sums_scalar(A, B, e, n, lambda, n_points, S);
}
void gkr_fold_batch(const GkrRec *A, const GkrRec *B, size_t n, const EF &r, GkrRec *out) {
#if defined(__x86_64__)
    if (n >= 4 && !scalar_only() && have_avx2()) return fold_avx2(A, B, n, r, out);
#endif
    fold_scalar(A, B, n, r, out);
}
void gkr_eq_batch(const EF *pt, unsigned nv, const uint64_t *y, size_t n, EF *out) {
#if defined(__x86_64__)
    if (n >= 4 && nv > 0 && !scalar_only() && have_avx2()) return eq_avx2(pt, nv, y, n, out);
#endif
    eq_scalar(pt, nv, y, n, out);
}

}  // namespace ovm

// 1 = the vectorised helpers and the scalar ones agree on `rounds` random batches of every size 1..40 (also on hosts
// without AVX2, where both are the scalar code)
extern "C" int ovm_hip_host_gkr_batch_selftest(uint32_t rounds, uint64_t seed) {
    using namespace ovm;
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        // (edge values included: 0, 1, P - 1)
        const uint32_t pick = (uint32_t)(st >> 60);
        const uint32_t canon = pick == 0 ? 0u : pick == 1 ? 1u : pick == 2 ? bb::P - 1 : (uint32_t)((st >> 20) % bb::P);
        return Fp::from_canonical(canon);
    };
    auto rnd_ef = [&]() {
        EF x;
        for (int k = 0; k < 4; k++) x.c[k] = rnd();
        return x;
    };
    for (uint32_t it = 0; it < rounds; it++)
        for (size_t n = 1; n <= 40; n++) {
            std::vector<GkrRec> A(n), B(n), f0(n), f1(n);
            std::vector<EF> e(n), q0(n), q1(n);
            std::vector<uint64_t> y(n);
            for (size_t i = 0; i < n; i++) {
                for (int k = 0; k < 4; k++) A[i].v[k] = rnd_ef(), B[i].v[k] = rnd_ef();
                e[i] = rnd_ef();
                y[i] = st >> 7;
                st = st * 6364136223846793005ull + 1;
            }
            const EF lambda = rnd_ef(), r = rnd_ef();
            EF pt[6];
            for (auto &p : pt) p = rnd_ef();
            const unsigned nv = (unsigned)(n % 7), n_points = 2 + (unsigned)(n & 1);
            EF s0[3] = {rnd_ef(), rnd_ef(), rnd_ef()}, s1[3] = {s0[0], s0[1], s0[2]};
            g_force_scalar = true;
            gkr_pair_sums_batch(A.data(), B.data(), e.data(), n, lambda, n_points, s0);
            gkr_fold_batch(A.data(), B.data(), n, r, f0.data());
            gkr_eq_batch(pt, nv > 6 ? 6 : nv, y.data(), n, q0.data());
            g_force_scalar = false;
            gkr_pair_sums_batch(A.data(), B.data(), e.data(), n, lambda, n_points, s1);
            gkr_fold_batch(A.data(), B.data(), n, r, f1.data());
            gkr_eq_batch(pt, nv > 6 ? 6 : nv, y.data(), n, q1.data());
            for (int k = 0; k < 3; k++)
                if (!(s0[k] == s1[k])) return 0;
            for (size_t i = 0; i < n; i++) {
                if (!(q0[i] == q1[i])) return 0;
                for (int k = 0; k < 4; k++)
                    if (!(f0[i].v[k] == f1[i].v[k])) return 0;
            }
        }
    return 1;
}
#endif
