// Host-side Poseidon2-BabyBear permutation for the Fiat-Shamir transcript: AVX2, the 16-word state in
// two 256-bit registers of Montgomery words. Bit-identical to p2::permute (poseidon2.h); selected at
// run time (__builtin_cpu_supports), so the library stays portable to hosts without AVX2.
//
// External layers: both halves through the S-box at once (x^7 = x^4 * x^3, four 8-lane Montgomery
// products), M4 on the four lanes of each 128-bit quarter by lane rotations (y_i = S + x_i + 2 x_{i+1},
// S = the quarter's sum), then the sum over the four quarters. Internal layers: the S-box of word 0 in
// scalar code, the sum of all 16 words by three rotate-and-add steps, the diagonal as one 8-lane
// product per half.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#ifndef OVM_HOST_FAST_PERMUTE
#define OVM_HOST_FAST_PERMUTE 1
#endif
#include "poseidon2.h"

#include <immintrin.h>

#include <algorithm>
#include <cstring>

namespace p2 {
namespace {

#define OVM_AVX2 __attribute__((target("avx2"))) inline

struct Tables {
    alignas(32) uint32_t diag[16];
    alignas(32) uint32_t rc_initial[64], rc_terminal[64];
    Tables() {
        // V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]
        auto inv2k = [](int k) {
            Fp x = Fp::one();
            for (int i = 0; i < k; i++) x = x.halve();
            return x;
        };
        const Fp one = Fp::one(), two = one + one, three = two + one, four = two + two;
        const Fp v[16] = {-two,        one,       two,       inv2k(1),  three,      four,       -inv2k(1),  -three,
                          -four,       inv2k(8),  inv2k(2),  inv2k(3),  inv2k(27),  -inv2k(8),  -inv2k(4),  -inv2k(27)};
        for (int i = 0; i < 16; i++) diag[i] = v[i].v;
        memcpy(rc_initial, RC_INITIAL, sizeof(rc_initial));
        memcpy(rc_terminal, RC_TERMINAL, sizeof(rc_terminal));
    }
};
const Tables &tables() {
    static const Tables t;
    return t;
}

OVM_AVX2 __m256i add_p(__m256i a, __m256i b) {
    const __m256i P8 = _mm256_set1_epi32((int)bb::P);
    const __m256i t = _mm256_add_epi32(a, b);
    return _mm256_min_epu32(t, _mm256_sub_epi32(t, P8));
}
// 8 Montgomery products: even and odd lanes as 64-bit products; x - (x * M mod 2^32) * P has a zero low
// word, its high word is xh - uh in (-P, P), corrected by min(r, r + P) as in bb::mont_mul
OVM_AVX2 __m256i mont_mul(__m256i a, __m256i b) {
    const __m256i P8 = _mm256_set1_epi32((int)bb::P), M8 = _mm256_set1_epi32((int)bb::MONTY_M);
    const __m256i a_odd = _mm256_srli_epi64(a, 32), b_odd = _mm256_srli_epi64(b, 32);
    const __m256i pe = _mm256_mul_epu32(a, b), po = _mm256_mul_epu32(a_odd, b_odd);
    const __m256i ue = _mm256_mul_epu32(_mm256_mul_epu32(pe, M8), P8), uo = _mm256_mul_epu32(_mm256_mul_epu32(po, M8), P8);
    const __m256i de = _mm256_sub_epi64(pe, ue), dodd = _mm256_sub_epi64(po, uo);
    const __m256i r = _mm256_blend_epi32(_mm256_srli_epi64(de, 32), dodd, 0xAA);
    return _mm256_min_epu32(r, _mm256_add_epi32(r, P8));
}
OVM_AVX2 __m256i sbox(__m256i x, __m256i rc) {
    const __m256i s = add_p(x, rc);
    const __m256i s2 = mont_mul(s, s), s3 = mont_mul(s2, s), s4 = mont_mul(s2, s2);
    return mont_mul(s4, s3);
}
OVM_AVX2 __m256i m4(__m256i x) {
    const __m256i r1 = _mm256_shuffle_epi32(x, 0x39), r2 = _mm256_shuffle_epi32(x, 0x4E), r3 = _mm256_shuffle_epi32(x, 0x93);
    const __m256i sum = add_p(add_p(x, r1), add_p(r2, r3));
    return add_p(add_p(sum, x), add_p(r1, r1));  // 2 x_i + 3 x_{i+1} + x_{i+2} + x_{i+3}
}
OVM_AVX2 void external_linear(__m256i &v0, __m256i &v1) {
    v0 = m4(v0);
    v1 = m4(v1);
    const __m256i t = add_p(v0, v1);
    const __m256i u = add_p(t, _mm256_permute2x128_si256(t, t, 0x01));
    v0 = add_p(v0, u);
    v1 = add_p(v1, u);
}
__attribute__((target("avx2"))) void permute_avx2(Fp *s) {
    const Tables &t = tables();
    __m256i v0 = _mm256_loadu_si256((const __m256i *)s), v1 = _mm256_loadu_si256((const __m256i *)(s + 8));
    const __m256i d0 = _mm256_load_si256((const __m256i *)t.diag), d1 = _mm256_load_si256((const __m256i *)(t.diag + 8));
    external_linear(v0, v1);
    for (int r = 0; r < 4; r++) {
        v0 = sbox(v0, _mm256_load_si256((const __m256i *)(t.rc_initial + 16 * r)));
        v1 = sbox(v1, _mm256_load_si256((const __m256i *)(t.rc_initial + 16 * r + 8)));
        external_linear(v0, v1);
    }
    // Internal layers. Word 0 lives in a scalar for the 13 rounds (lane 0 of v0 is kept at zero): its S-box -- three
    // dependent Montgomery products, the longest chain of the round -- then runs BESIDE the vector work that does not
    // need it (the sum of words 1..15 and the diagonal products), and the round's critical path is the scalar chain
    // alone:  s0' = part - y,  s_i' = d_i s_i + (part + y)  with  y = (s0 + rc)^7,  part = s_1 + ... + s_15.
    const __m256i zero = _mm256_setzero_si256();
    Fp x0 = Fp::raw((uint32_t)_mm256_cvtsi256_si32(v0));
    v0 = _mm256_blend_epi32(v0, zero, 0x01);
    for (int r = 0; r < 13; r++) {
        const Fp y = sbox7_add(x0, RC_INTERNAL[r]);
        __m256i part = add_p(v0, v1);
        part = add_p(part, _mm256_permute2x128_si256(part, part, 0x01));
        part = add_p(part, _mm256_shuffle_epi32(part, 0x4E));
        part = add_p(part, _mm256_shuffle_epi32(part, 0xB1));  // every lane: s_1 + ... + s_15
        const __m256i p0 = mont_mul(v0, d0), p1 = mont_mul(v1, d1);  // (lane 0 of p0 stays zero)
        const Fp part_s = Fp::raw((uint32_t)_mm256_cvtsi256_si32(part));
        x0 = part_s - y;
        const __m256i sum = _mm256_set1_epi32((int)(part_s + y).v);
        v0 = _mm256_blend_epi32(add_p(p0, sum), zero, 0x01);
        v1 = add_p(p1, sum);
    }
    v0 = _mm256_blend_epi32(v0, _mm256_castsi128_si256(_mm_cvtsi32_si128((int)x0.v)), 0x01);
    for (int r = 0; r < 4; r++) {
        v0 = sbox(v0, _mm256_load_si256((const __m256i *)(t.rc_terminal + 16 * r)));
        v1 = sbox(v1, _mm256_load_si256((const __m256i *)(t.rc_terminal + 16 * r + 8)));
        external_linear(v0, v1);
    }
    _mm256_storeu_si256((__m256i *)s, v0);
    _mm256_storeu_si256((__m256i *)(s + 8), v1);
}

// ---- eight permutations at once (one state per 32-bit lane, word i of all eight in v[i]) -----------------------------
// For the proof-of-work searches the transcript runs on the host (folding_pow_bits = 5: ~32 candidate witnesses per
// WHIR folding round, 16 rounds per proof): the candidates are independent, so eight of them share every instruction --
// ~0.25 us per candidate instead of ~1 us. The same field operations as p2::permute on canonical Montgomery words.
OVM_AVX2 __m256i dbl_p(__m256i a) { return add_p(a, a); }
OVM_AVX2 void mat4_x8(__m256i *x) {
    const __m256i t01 = add_p(x[0], x[1]), t23 = add_p(x[2], x[3]);
    const __m256i t0123 = add_p(t01, t23);
    const __m256i t01123 = add_p(t0123, x[1]), t01233 = add_p(t0123, x[3]);
    x[3] = add_p(t01233, dbl_p(x[0]));
    x[1] = add_p(t01123, dbl_p(x[2]));
    x[0] = add_p(t01123, t01);
    x[2] = add_p(t01233, t23);
}
OVM_AVX2 void external_linear_x8(__m256i *v) {
    for (int b = 0; b < 4; b++) mat4_x8(v + 4 * b);
    __m256i sums[4];
    for (int j = 0; j < 4; j++) sums[j] = add_p(add_p(v[j], v[4 + j]), add_p(v[8 + j], v[12 + j]));
    for (int i = 0; i < 16; i++) v[i] = add_p(v[i], sums[i & 3]);
}
__attribute__((target("avx2"))) void permute_x8(__m256i *v) {
    const Tables &t = tables();
    external_linear_x8(v);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) v[i] = sbox(v[i], _mm256_set1_epi32((int)t.rc_initial[16 * r + i]));
        external_linear_x8(v);
    }
    for (int r = 0; r < 13; r++) {
        v[0] = sbox(v[0], _mm256_set1_epi32((int)RC_INTERNAL[r]));
        __m256i sum = v[0];
        for (int i = 1; i < 16; i++) sum = add_p(sum, v[i]);
        for (int i = 0; i < 16; i++) v[i] = add_p(mont_mul(v[i], _mm256_set1_epi32((int)t.diag[i])), sum);  // (J + diag(V)) v
    }
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) v[i] = sbox(v[i], _mm256_set1_epi32((int)t.rc_terminal[16 * r + i]));
        external_linear_x8(v);
    }
}
// the smallest w in [from, from + 8) whose sponge -- s after observe(w) -- samples `bits` zero bits, or ~0
__attribute__((target("avx2"))) uint32_t grind_scan8(const Sponge &s, unsigned bits, uint32_t from) {
    __m256i v[16];
    for (int i = 0; i < 16; i++) v[i] = _mm256_set1_epi32((int)s.state[i].v);
    alignas(32) uint32_t w[8];
    for (int l = 0; l < 8; l++) w[l] = Fp::from_canonical(from + l).v;
    v[s.absorb_idx] = _mm256_load_si256((const __m256i *)w);
    permute_x8(v);  // (observe permutes when the witness fills the rate, sample when it does not: one permutation either way)
    alignas(32) uint32_t out[8];
    _mm256_store_si256((__m256i *)out, v[RATE - 1]);  // the first sample after a permutation is word 7
    for (int l = 0; l < 8; l++)
        if ((Fp::raw(out[l]).canonical() & ((1u << bits) - 1)) == 0) return from + l;
    return ~0u;
}

}  // namespace

// The minimum witness of a proof-of-work of `bits` bits on sponge s (what the scalar loop "observe w, sample_bits" finds):
// eight candidates per step where the host has AVX2.
uint32_t host_grind_min_witness(const Sponge &s, unsigned bits) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) {
        for (uint32_t from = 0; from + 8 <= bb::P; from += 8) {
            const uint32_t w = grind_scan8(s, bits, from);
            if (w != ~0u) return w;
        }
        return ~0u;
    }
    for (uint32_t w = 0; w < bb::P; w++) {
        Sponge t = s;
        t.observe(Fp::from_canonical(w));
        if (t.sample_bits(bits) == 0) return w;
    }
    return ~0u;
}

void host_permute(Fp *s) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) permute_avx2(s);
    else permute(s);
}

}  // namespace p2

// Worst-case ranges of the lazily reduced permutation (poseidon2.h: lazy::permute), re-derived from the ACTUAL constants
// of its tables instead of the "|K| <= P/2" of the header comment: the magnitude of every 32-bit word and 64-bit sum is
// propagated through the rounds for the worst input (every canonical word may be P - 1, every sign may conspire), and every
// place where the code truncates to int32, multiplies two words or reduces a sum is checked against what that operation can
// hold. false = some bound of the comment does not hold for these tables.
static bool lazy_range_audit() {
    using namespace p2::lazy;
    const long double P = (long double)bb::P, H = 4294967296.0L, I32 = 2147483648.0L, REDC_MAX = 4611686018427387904.0L;  // 2^62
    auto mag = [](int64_t v) { return (long double)(v < 0 ? -v : v); };
    bool ok = true;
    auto need = [&](bool c) { ok = ok && c; };
    auto redc = [&](long double t) {  // |result| of smont_redc on a sum of magnitude t
        need(t <= REDC_MAX);
        return t / H + P / 2 + 1;
    };
    auto prod = [&](long double a, long double b) { return redc(a * b); };  // smont_mul: both words fit int32, the product 2^62
    auto sbox = [&](long double s) {
        need(s < I32);
        const long double s2 = prod(s, s), s3 = prod(s2, s), s4 = prod(s2, s2);
        need(s2 < I32 && s3 < I32 && s4 < I32);
        const long double s7 = prod(s4, s3);
        need(s7 < P);  // (the comment's .921 P / .95 P; what matters: it is a word)
        return s7;
    };
    auto external = [&](long double b) {  // 16 words of magnitude b -> the reduced words
        const long double y = 7 * (5 * b);
        need(y < 9.0e18L);  // the 64-bit sums themselves
        return redc(y);
    };
    auto maxabs = [&](const int32_t *t, int n) {
        long double m = 0;
        for (int i = 0; i < n; i++) m = std::max(m, mag(t[i]));
        return m;
    };
    long double b = external(P - 1);  // the first layer on canonical words
    for (int r = 0; r < 4; r++) b = external(sbox(b + maxabs(LAZY.c_initial + 16 * r, 16)));
    // first internal round
    const long double SC = (8.0L / 15.0L) * P + 32;  // centred_sum: |S - q P| < 8 * 2^27 + |q|
    {
        const long double x0 = sbox(b + mag(LAZY.c_internal0));
        need(15 * b / 134217728.0L < 131072.0L);  // S >> 27 times 4369 stays in int32
        const long double U = SC * mag(LAZY.first.k_sum), T = x0 * mag(LAZY.first.k_word0) + U;
        long double w = 0;
        for (int i = 1; i < 16; i++) w = std::max(w, redc(b * mag(LAZY.first.diag[i]) + T));
        const long double w0 = redc(x0 * mag(LAZY.first.diag[0]) + U + P);  // (+ the round constant riding along, < P)
        need(w < I32 && w0 < I32);
        b = w;
        // middle rounds and the last one: word 0 is the S-box input w0
        long double in0 = w0;
        for (int r = 1; r < 13; r++) {
            const RoundK &K = r < 12 ? LAZY.middle : LAZY.last;
            const long double y0 = sbox(in0);
            need((y0 + 15 * b) / 134217728.0L < 131072.0L);
            const long double Tm = SC * mag(K.k_sum);
            long double wn = 0;
            for (int i = 1; i < 16; i++) wn = std::max(wn, redc(b * mag(K.diag[i]) + Tm + (r == 12 ? P : 0)));
            const long double wn0 = redc(y0 * mag(K.diag[0]) + Tm + P);
            need(wn < I32 && wn0 < I32);
            b = wn, in0 = wn0;
        }
        b = std::max(b, in0);  // the sixteen inputs of the first terminal S-boxes
    }
    b = external(sbox(b));  // terminal round 0: its constants rode in the last internal round
    for (int r = 1; r < 4; r++) b = external(sbox(b + maxabs(LAZY.c_terminal + 16 * r, 16)));
    need(b < P);  // from_signed: a representative in (-P, P)
    return ok;
}

// Self-test hook (tests/test_abi.py): 1 if the dispatcher and the scalar permutation agree on n states
extern "C" int ovm_hip_host_permute_selftest(uint32_t n, uint64_t seed) {
    if (!lazy_range_audit()) return 0;
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
    for (uint32_t it = 0; it < n; it++) {
        p2::Fp a[16], b[16];
        for (int i = 0; i < 16; i++) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            a[i] = b[i] = p2::Fp::from_canonical((uint32_t)((st >> 33) % bb::P));
        }
        if (it < 64) {  // edge states first: all zero, all P - 1, single extreme words, alternating extremes
            for (int i = 0; i < 16; i++) {
                const uint32_t hi = bb::P - 1;
                a[i] = b[i] = p2::Fp::raw(it == 0 ? 0 : it == 1 ? hi : it < 18 ? (i == (int)it - 2 ? hi : 0) : it < 34 ? (i == (int)it - 18 ? 0 : hi)
                                          : ((it >> (i & 3)) & 1) ? hi : (it & 32) ? bb::P / 2 + i : 0);
            }
        }
        // the lazily reduced form the kernels run (p2::permute_lazy_t and its first / last layer variants, and the form
        // that scales round constants as it reads them) against the plain statement of the layers
        {
            p2::Fp l[16], r[16];
            for (int i = 0; i < 16; i++) l[i] = r[i] = a[i];
            p2::permute_lazy_t<16, false>(l);
            p2::permute_ref_t<16, false>(r);
            for (int i = 0; i < 16; i++)
                if (l[i].v != r[i].v) return 0;
            for (int i = 0; i < 16; i++) l[i] = r[i] = a[i];
            p2::permute_lazy_t<16, true>(l);
            p2::permute_ref_t<16, true>(r);
            for (int i = 0; i < 8; i++)
                if (l[i].v != r[i].v) return 0;
            for (int i = 0; i < 16; i++) l[i] = r[i] = i < 8 ? a[i] : p2::Fp::zero();
            p2::permute_lazy_t<8, false>(l);
            p2::permute_ref_t<8, false>(r);
            for (int i = 0; i < 16; i++)
                if (l[i].v != r[i].v) return 0;
            for (int i = 0; i < 16; i++) l[i] = r[i] = i < 8 ? a[i] : p2::Fp::zero();
            p2::permute_lazy_t<8, true>(l);
            p2::permute_ref_t<8, true>(r);
            for (int i = 0; i < 8; i++)
                if (l[i].v != r[i].v) return 0;
            for (int i = 0; i < 16; i++) l[i] = r[i] = i < 4 ? a[i] : p2::Fp::zero();
            p2::permute_lazy_t<4, true>(l);
            p2::permute_ref_t<4, true>(r);
            for (int i = 0; i < 8; i++)
                if (l[i].v != r[i].v) return 0;
            uint32_t rc[141];
            for (int i = 0; i < 64; i++) rc[i] = p2::RC_INITIAL[i], rc[64 + i] = p2::RC_TERMINAL[i];
            for (int i = 0; i < 13; i++) rc[128 + i] = p2::RC_INTERNAL[i];
            if (it & 1)  // (and with constants that are not the table's)
                for (int i = 0; i < 141; i++) rc[i] = (uint32_t)(((uint64_t)rc[i] * 2654435761u + st) % bb::P);
            for (int i = 0; i < 16; i++) l[i] = r[i] = a[i];
            p2::lazy::permute<p2::lazy::MemoryRC, 16, false>(l, p2::lazy::MemoryRC{rc});
            p2::permute_rc(r, rc);
            for (int i = 0; i < 16; i++)
                if (l[i].v != r[i].v) return 0;
        }
        p2::permute(a);
        p2::host_permute(b);
        for (int i = 0; i < 16; i++)
            if (a[i].v != b[i].v) return 0;
        // the eight-candidate search against the scalar one, at every absorb position the transcript can be in
        p2::Sponge sp;
        for (int i = 0; i < 16; i++) sp.state[i] = a[i];
        sp.absorb_idx = it % 8;
        sp.sample_idx = (it / 8) % 9;
        const unsigned bits = 1 + it % 7;
        uint32_t want = ~0u;
        for (uint32_t w = 0; w < bb::P && want == ~0u; w++) {
            p2::Sponge t = sp;
            t.observe(p2::Fp::from_canonical(w));
            if (t.sample_bits(bits) == 0) want = w;
        }
        if (p2::host_grind_min_witness(sp, bits) != want) return 0;
    }
    return 1;
}
#endif
