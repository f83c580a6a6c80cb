// BabyBear (p = 15*2^27 + 1) in 32-bit Montgomery form (R = 2^32) and its quartic extension
// F[X]/(X^4 - 11), usable from host C++ and gfx950 device code.
//
// The in-memory representation is the one the reference passes across its FFI unchanged: a Rust
// `BabyBear` is a `repr(transparent)` Montgomery u32, an extension element is `[u32; 4]`
// (reference: crates/cuda-common/include/fp.h:52-60,184-194 and fpext.h:28-44).
//
// CDNA4 notes (measured on MI355X with tools/ubench_int64.hip, round 6): v_mad_u64_u32 / v_mad_i64_i32 and
// v_lshl_add_u64 issue at the rate of v_mul_lo_u32 / v_mul_hi_u32 / v_min_u32 (~4.3-4.7 cycles per wave-instruction
// per SIMD; v_add_u32, v_sub_u32 and the 32-bit shifts ~2.5-2.9). Rounds 1-5 priced the 64-bit multiply-add at two
// slots from a microbenchmark that had an uncounted v_xor in its loop body; it costs ONE. So a Montgomery product is
// THREE instructions: the 64-bit product (mad), lo * P^-1 (mul_lo), and product - m * P as a second 64-bit mad whose
// high word is the result -- instead of two mul_lo, two mul_hi and a subtract. The compiler selects exactly these
// from the 64-bit C++ expressions below.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BB_HD __host__ __device__ __forceinline__
#else
#define BB_HD inline
#endif

namespace bb {

static constexpr uint32_t P = 0x78000001u;
static constexpr uint32_t MONTY_M = 0x88000001u;  // -P^-1 ... P * M == 1 (mod 2^32), see fp.h:56
static constexpr uint32_t R2 = 1172168163u;       // 2^64 mod P

// ---- compile-time helpers (used to bake constants in Montgomery form) ----
constexpr uint32_t cx_to_mont(uint64_t x) { return (uint32_t)(((x % P) << 32) % P); }
constexpr uint32_t cx_mulmod(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); }
constexpr uint32_t cx_powmod(uint32_t a, uint64_t e) {
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = cx_mulmod(r, a);
        a = cx_mulmod(a, a);
        e >>= 1;
    }
    return r;
}

BB_HD uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

static constexpr uint32_t MONTY_NEG_M = 0u - MONTY_M;  // -P^-1 mod 2^32
static constexpr int32_t NEG_P = -(int32_t)P;

// x < 2^32 * P  ->  x * 2^-32 mod P, in [0, P).
//   m = lo(x) * (-P^-1) mod 2^32 ; x + m * P has a zero low word and is < 2^33 P < 2^64, so its high word is
//   hi(x) + hi(m * P) (+ the carry), in [0, 2P): one 64-bit multiply-add, then `min(r, r - P)`.
BB_HD uint32_t mont_reduce(uint64_t x) {
    const uint32_t m = (uint32_t)x * MONTY_NEG_M;
    const uint32_t r = (uint32_t)((x + (uint64_t)m * P) >> 32);
    return umin32(r, r - P);
}

// Same, for x < 4*P^2 (a sum of up to four raw products): hi(x) may reach ~1.9 P, so it is
// brought below P first (sub + min), after which the sum above cannot overflow.
BB_HD uint32_t mont_reduce_wide(uint64_t x) {
    uint32_t xh = (uint32_t)(x >> 32);
    xh = umin32(xh, xh - P);
    const uint32_t m = (uint32_t)x * MONTY_NEG_M;
    const uint32_t r = (uint32_t)(((((uint64_t)xh) << 32 | (uint32_t)x) + (uint64_t)m * P) >> 32);
    return umin32(r, r - P);
}

// Signed Montgomery product: a * b * 2^-32 mod P as a representative in (-P, P), for int32
// |a| <= P, |b| < P. x = a*b and m*P (m = lo(x) * P^-1 as a SIGNED word) have equal low words, so
// (x - m*P) >> 32 is exact; |hi(x)| < 0.47 P and |hi(m*P)| <= P/2 bound the result without any correction
// step. Chains of multiplies (the Poseidon2 S-box) run on these and canonicalise once at the end.
// In fact any int32 a, b are admissible: |hi(x)| <= 2^30 and |hi(m*P)| <= P/2 keep the result inside int32.
BB_HD int32_t smont_redc(int64_t x) {
    const int32_t m = (int32_t)((uint32_t)x * MONTY_M);
    return (int32_t)((x + (int64_t)m * (int64_t)NEG_P) >> 32);
}
BB_HD int32_t smont_mul(int32_t a, int32_t b) { return smont_redc((int64_t)a * (int64_t)b); }
// representative in (-P, P) -> canonical word
BB_HD uint32_t from_signed(int32_t r) { return umin32((uint32_t)r, (uint32_t)r + P); }

// a * b * 2^-32 mod P for a, b < P: the signed product (hi(a*b) < 0.47 P keeps it in (-P/2, P)), canonicalised.
BB_HD uint32_t mont_mul(uint32_t a, uint32_t b) { return from_signed(smont_redc((int64_t)((uint64_t)a * b))); }

struct Fp {
    uint32_t v;  // Montgomery word, < P

    BB_HD static Fp raw(uint32_t m) {
        Fp r;
        r.v = m;
        return r;
    }
    BB_HD static Fp zero() { return raw(0); }
    BB_HD static Fp one() { return raw(0x0ffffffeu); }  // 2^32 mod P
    BB_HD static Fp from_canonical(uint32_t x) { return raw(mont_reduce((uint64_t)x * R2)); }
    BB_HD uint32_t canonical() const { return mont_reduce((uint64_t)v); }

    BB_HD Fp operator+(Fp o) const {
        uint32_t s = v + o.v;  // < 2P < 2^32
        return raw(umin32(s, s - P));
    }
    BB_HD Fp operator-(Fp o) const {
        uint32_t d = v - o.v;  // wraps iff negative
        return raw(umin32(d, d + P));
    }
    BB_HD Fp operator-() const { return raw(v ? P - v : 0); }
    BB_HD Fp operator*(Fp o) const { return raw(mont_mul(v, o.v)); }
    BB_HD Fp &operator+=(Fp o) { return *this = *this + o; }
    BB_HD Fp &operator-=(Fp o) { return *this = *this - o; }
    BB_HD Fp &operator*=(Fp o) { return *this = *this * o; }
    BB_HD bool operator==(Fp o) const { return v == o.v; }
    BB_HD bool operator!=(Fp o) const { return v != o.v; }
    BB_HD Fp dbl() const { return *this + *this; }
    BB_HD Fp square() const { return *this * *this; }
    BB_HD Fp halve() const { return raw((v >> 1) + ((v & 1) ? (P + 1) / 2 : 0)); }
    BB_HD Fp pow(uint64_t e) const {
        Fp b = *this, r = one();
        while (e) {
            if (e & 1) r *= b;
            b *= b;
            e >>= 1;
        }
        return r;
    }
    BB_HD Fp exp_power_of_2(unsigned k) const {
        Fp r = *this;
        for (unsigned i = 0; i < k; i++) r = r.square();
        return r;
    }
    BB_HD Fp inv() const { return pow(P - 2); }
};

// Compile-time Montgomery constant.
template <uint32_t CANON>
BB_HD Fp fp_const() {
    return Fp::raw(cx_to_mont(CANON));
}

// reference: crates/cuda-common/include/fp.h:291-320 (canonical values)
static constexpr uint32_t TWO_ADIC_GENERATORS[28] = {
    0x1,        0x78000000, 0x67055c21, 0x5ee99486, 0xbb4c4e4,  0x2d4cc4da, 0x669d6090,
    0x17b56c64, 0x67456167, 0x688442f9, 0x145e952d, 0x4fe61226, 0x4c734715, 0x11c33e2a,
    0x62c3d2b1, 0x77cad399, 0x54c131f4, 0x4cabd6a6, 0x5cf5713f, 0x3e9430e8, 0xba067a3,
    0x18adc27d, 0x21fd55bc, 0x4b859b3d, 0x3bd57996, 0x4483d85a, 0x3a26eef8, 0x1a427a41};
static constexpr uint32_t GENERATOR = 31;  // p3-baby-bear: multiplicative generator of F*

inline Fp two_adic_generator(unsigned bits) {
    return Fp::from_canonical(TWO_ADIC_GENERATORS[bits]);
}

struct alignas(4) FpExt {
    Fp c[4];

    BB_HD static FpExt zero() {
        FpExt r;
        r.c[0] = r.c[1] = r.c[2] = r.c[3] = Fp::zero();
        return r;
    }
    BB_HD static FpExt one() {
        FpExt r = zero();
        r.c[0] = Fp::one();
        return r;
    }
    BB_HD static FpExt from_base(Fp a) {
        FpExt r = zero();
        r.c[0] = a;
        return r;
    }
    BB_HD FpExt operator+(const FpExt &o) const {
        FpExt r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[i] = c[i] + o.c[i];
        return r;
    }
    BB_HD FpExt operator-(const FpExt &o) const {
        FpExt r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[i] = c[i] - o.c[i];
        return r;
    }
    BB_HD FpExt operator-() const {
        FpExt r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[i] = -c[i];
        return r;
    }
    BB_HD FpExt operator*(Fp s) const {
        FpExt r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[i] = c[i] * s;
        return r;
    }
    BB_HD FpExt operator+(Fp s) const {
        FpExt r = *this;
        r.c[0] += s;
        return r;
    }
    BB_HD FpExt operator-(Fp s) const {
        FpExt r = *this;
        r.c[0] -= s;
        return r;
    }
    // Schoolbook product mod X^4 = 11 with the wrap-around factor folded into the right operand:
    // with w_k = 11 b_k every coefficient is ONE 64-bit sum of four raw products (< 4 P^2 < 2^64)
    // and one Montgomery reduction -- 16 multiply-adds and 4 reductions. The three w_k cost a
    // multiply each, paid once per right operand: kernels that multiply many values by the same
    // challenge (folds by r, lambda, eq weights) precompute them (Pre).
    struct Pre {
        Fp b[4];
        Fp w[3];  // 11 * b[1..3]
    };
    BB_HD Pre pre() const {
        const Fp W = fp_const<11>();
        Pre p;
#pragma unroll
        for (int i = 0; i < 4; i++) p.b[i] = c[i];
#pragma unroll
        for (int i = 0; i < 3; i++) p.w[i] = c[i + 1] * W;
        return p;
    }
    BB_HD FpExt operator*(const Pre &o) const {
        const Fp *a = c;
        const Fp *b = o.b, *w = o.w - 1;  // w[k] = 11 b[k], k = 1..3
        FpExt r;
        r.c[0] = Fp::raw(mont_reduce_wide((uint64_t)a[0].v * b[0].v + (uint64_t)a[1].v * w[3].v +
                                          (uint64_t)a[2].v * w[2].v + (uint64_t)a[3].v * w[1].v));
        r.c[1] = Fp::raw(mont_reduce_wide((uint64_t)a[0].v * b[1].v + (uint64_t)a[1].v * b[0].v +
                                          (uint64_t)a[2].v * w[3].v + (uint64_t)a[3].v * w[2].v));
        r.c[2] = Fp::raw(mont_reduce_wide((uint64_t)a[0].v * b[2].v + (uint64_t)a[1].v * b[1].v +
                                          (uint64_t)a[2].v * b[0].v + (uint64_t)a[3].v * w[3].v));
        r.c[3] = Fp::raw(mont_reduce_wide((uint64_t)a[0].v * b[3].v + (uint64_t)a[1].v * b[2].v +
                                          (uint64_t)a[2].v * b[1].v + (uint64_t)a[3].v * b[0].v));
        return r;
    }
    BB_HD FpExt operator*(const FpExt &o) const { return *this * o.pre(); }
    BB_HD FpExt &operator+=(const FpExt &o) { return *this = *this + o; }
    BB_HD FpExt &operator-=(const FpExt &o) { return *this = *this - o; }
    BB_HD FpExt &operator*=(const FpExt &o) { return *this = *this * o; }
    BB_HD FpExt &operator*=(Fp o) { return *this = *this * o; }
    BB_HD bool operator==(const FpExt &o) const {
        return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2] && c[3] == o.c[3];
    }
    BB_HD bool operator!=(const FpExt &o) const { return !(*this == o); }
    BB_HD FpExt dbl() const { return *this + *this; }
    BB_HD FpExt square() const { return *this * *this; }
    BB_HD FpExt halve() const {
        FpExt r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[i] = c[i].halve();
        return r;
    }
    BB_HD FpExt pow(uint64_t e) const {
        FpExt b = *this, r = one();
        while (e) {
            if (e & 1) r *= b;
            b *= b;
            e >>= 1;
        }
        return r;
    }
    BB_HD FpExt exp_power_of_2(unsigned k) const {
        FpExt r = *this;
        for (unsigned i = 0; i < k; i++) r = r.square();
        return r;
    }
    BB_HD FpExt inv() const {
        // Tower: a = A + X*B over F[Y]/(Y^2 - 11), Y = X^2; a^-1 = (A - X*B) / (A^2 - Y*B^2).
        const Fp W = fp_const<11>();
        Fp a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
        Fp A2_0 = a0 * a0 + W * (a2 * a2), A2_1 = (a0 * a2).dbl();
        Fp B2_0 = a1 * a1 + W * (a3 * a3), B2_1 = (a1 * a3).dbl();
        Fp c0 = A2_0 - W * B2_1, c1 = A2_1 - B2_0;
        Fp n = (c0 * c0 - W * (c1 * c1)).inv();
        Fp d0 = c0 * n, d1 = -(c1 * n);
        FpExt r;
        r.c[0] = a0 * d0 + W * (a2 * d1);
        r.c[2] = a0 * d1 + a2 * d0;
        r.c[1] = -(a1 * d0 + W * (a3 * d1));
        r.c[3] = -(a1 * d1 + a3 * d0);
        return r;
    }
};

BB_HD FpExt operator*(Fp a, const FpExt &b) { return b * a; }

}  // namespace bb
