// ORACLE — test infrastructure only (see oracle/README.md). Never linked into the product library.
//
// BabyBear base field and its quartic extension, in CANONICAL (non-Montgomery) form with plain
// 64-bit `%` reduction. Deliberately the slow, obviously-correct formulation: the product
// (stark-backend_amd/csrc/bb31.h) uses 32-bit Montgomery form like the reference, so agreement
// between the two is a cross-implementation check.
//
// Follows (reference, relative to /root/reference):
//   crates/cuda-common/include/fp.h:55-59      P = 15*2^27+1, Montgomery constants (boundary only)
//   crates/cuda-common/include/fp.h:291-320    TWO_ADIC_GENERATORS[0..27]
//   crates/cuda-common/include/fpext.h:28-44   EF = F[X]/(X^4 - 11), basis 1,X,X^2,X^3
//   p3-baby-bear =0.4.3 (absent from the tree): multiplicative generator GENERATOR = 31
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <vector>

namespace oracle {

static constexpr uint32_t P = 2013265921u;  // 15 * 2^27 + 1

struct Fp {
    uint32_t v;  // canonical, < P
    constexpr Fp() : v(0) {}
    constexpr explicit Fp(uint64_t x) : v((uint32_t)(x % P)) {}
    static constexpr Fp zero() { return Fp(); }
    static constexpr Fp one() { return Fp(1); }
    static Fp from_i64(int64_t x) {
        int64_t r = x % (int64_t)P;
        if (r < 0) r += P;
        return Fp((uint64_t)r);
    }
    Fp operator+(Fp o) const {
        uint32_t s = v + o.v;
        Fp r;
        r.v = s >= P ? s - P : s;
        return r;
    }
    Fp operator-(Fp o) const {
        Fp r;
        r.v = v >= o.v ? v - o.v : v + P - o.v;
        return r;
    }
    Fp operator-() const {
        Fp r;
        r.v = v ? P - v : 0;
        return r;
    }
    Fp operator*(Fp o) const {
        Fp r;
        r.v = (uint32_t)(((uint64_t)v * o.v) % P);
        return r;
    }
    Fp &operator+=(Fp o) { return *this = *this + o; }
    Fp &operator-=(Fp o) { return *this = *this - o; }
    Fp &operator*=(Fp o) { return *this = *this * o; }
    bool operator==(Fp o) const { return v == o.v; }
    bool operator!=(Fp o) const { return v != o.v; }
    Fp dbl() const { return *this + *this; }
    Fp square() const { return *this * *this; }
    Fp pow(uint64_t e) const {
        Fp b = *this, r = one();
        while (e) {
            if (e & 1) r *= b;
            b *= b;
            e >>= 1;
        }
        return r;
    }
    Fp exp_power_of_2(unsigned k) const {
        Fp r = *this;
        for (unsigned i = 0; i < k; i++) r = r.square();
        return r;
    }
    Fp inv() const {
        assert(v != 0);
        return pow(P - 2);
    }
    Fp halve() const { return *this * Fp((uint64_t)(P + 1) / 2); }
    // Montgomery boundary (R = 2^32): what a Rust `BabyBear` holds in memory.
    static Fp from_mont(uint32_t m) {
        // m = x * 2^32 mod P  =>  x = m * 2^-32 mod P
        static const Fp rinv = Fp((uint64_t)1 << 32).inv();
        return Fp((uint64_t)m) * rinv;
    }
    uint32_t to_mont() const { return (uint32_t)((((uint64_t)v) << 32) % P); }
};

static const Fp GENERATOR = Fp(31);

// cuda-common/include/fp.h:291-320 (canonical values)
static const uint32_t TWO_ADIC_GENERATORS[28] = {
    0x1,        0x78000000, 0x67055c21, 0x5ee99486, 0xbb4c4e4,  0x2d4cc4da, 0x669d6090,
    0x17b56c64, 0x67456167, 0x688442f9, 0x145e952d, 0x4fe61226, 0x4c734715, 0x11c33e2a,
    0x62c3d2b1, 0x77cad399, 0x54c131f4, 0x4cabd6a6, 0x5cf5713f, 0x3e9430e8, 0xba067a3,
    0x18adc27d, 0x21fd55bc, 0x4b859b3d, 0x3bd57996, 0x4483d85a, 0x3a26eef8, 0x1a427a41};

inline Fp two_adic_generator(unsigned bits) {
    assert(bits <= 27);
    return Fp((uint64_t)TWO_ADIC_GENERATORS[bits]);
}

struct EF {
    Fp c[4];
    EF() {}
    explicit EF(Fp a) { c[0] = a; }
    EF(Fp a, Fp b, Fp cc, Fp d) {
        c[0] = a;
        c[1] = b;
        c[2] = cc;
        c[3] = d;
    }
    static EF zero() { return EF(); }
    static EF one() { return EF(Fp::one()); }
    static EF from_u64(uint64_t x) { return EF(Fp(x)); }
    EF operator+(const EF &o) const {
        EF r;
        for (int i = 0; i < 4; i++) r.c[i] = c[i] + o.c[i];
        return r;
    }
    EF operator-(const EF &o) const {
        EF r;
        for (int i = 0; i < 4; i++) r.c[i] = c[i] - o.c[i];
        return r;
    }
    EF operator-() const {
        EF r;
        for (int i = 0; i < 4; i++) r.c[i] = -c[i];
        return r;
    }
    EF operator*(Fp s) const {
        EF r;
        for (int i = 0; i < 4; i++) r.c[i] = c[i] * s;
        return r;
    }
    EF operator+(Fp s) const {
        EF r = *this;
        r.c[0] += s;
        return r;
    }
    EF operator-(Fp s) const {
        EF r = *this;
        r.c[0] -= s;
        return r;
    }
    EF operator*(const EF &o) const {
        // schoolbook mod X^4 = 11
        static const Fp W = Fp(11);
        const Fp *a = c, *b = o.c;
        EF r;
        r.c[0] = a[0] * b[0] + W * (a[1] * b[3] + a[2] * b[2] + a[3] * b[1]);
        r.c[1] = a[0] * b[1] + a[1] * b[0] + W * (a[2] * b[3] + a[3] * b[2]);
        r.c[2] = a[0] * b[2] + a[1] * b[1] + a[2] * b[0] + W * (a[3] * b[3]);
        r.c[3] = a[0] * b[3] + a[1] * b[2] + a[2] * b[1] + a[3] * b[0];
        return r;
    }
    EF &operator+=(const EF &o) { return *this = *this + o; }
    EF &operator-=(const EF &o) { return *this = *this - o; }
    EF &operator*=(const EF &o) { return *this = *this * o; }
    EF &operator*=(Fp o) { return *this = *this * o; }
    bool operator==(const EF &o) const {
        return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2] && c[3] == o.c[3];
    }
    bool operator!=(const EF &o) const { return !(*this == o); }
    bool is_zero() const { return *this == EF(); }
    EF dbl() const { return *this + *this; }
    EF square() const { return *this * *this; }
    EF halve() const { return *this * Fp::one().halve(); }
    EF pow(uint64_t e) const {
        EF b = *this, r = one();
        while (e) {
            if (e & 1) r *= b;
            b *= b;
            e >>= 1;
        }
        return r;
    }
    EF exp_power_of_2(unsigned k) const {
        EF r = *this;
        for (unsigned i = 0; i < k; i++) r = r.square();
        return r;
    }
    EF inv() const {
        // a = A + X*B with A = a0 + a2*Y, B = a1 + a3*Y, Y = X^2, Y^2 = 11.
        // a * (A - X*B) = A^2 - Y*B^2 =: C = c0 + c1*Y ; C^-1 = (c0 - c1*Y) / (c0^2 - 11 c1^2)
        static const Fp W = Fp(11);
        Fp a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
        // A^2 = (a0^2 + 11 a2^2) + (2 a0 a2) Y ; B^2 = (a1^2 + 11 a3^2) + (2 a1 a3) Y
        Fp A2_0 = a0 * a0 + W * a2 * a2, A2_1 = (a0 * a2).dbl();
        Fp B2_0 = a1 * a1 + W * a3 * a3, B2_1 = (a1 * a3).dbl();
        // Y*B^2 = 11*B2_1 + B2_0*Y
        Fp c0 = A2_0 - W * B2_1, c1 = A2_1 - B2_0;
        Fp n = (c0 * c0 - W * c1 * c1).inv();
        Fp d0 = c0 * n, d1 = -(c1 * n);  // C^-1 = d0 + d1*Y
        // (A - X*B) * (d0 + d1*Y), with Y = X^2:
        //   A*Cinv = (a0 d0 + 11 a2 d1) + (a0 d1 + a2 d0) Y
        //   B*Cinv = (a1 d0 + 11 a3 d1) + (a1 d1 + a3 d0) Y
        EF r;
        r.c[0] = a0 * d0 + W * a2 * d1;
        r.c[2] = a0 * d1 + a2 * d0;
        r.c[1] = -(a1 * d0 + W * a3 * d1);
        r.c[3] = -(a1 * d1 + a3 * d0);
        return r;
    }
};

inline EF operator*(Fp a, const EF &b) { return b * a; }

template <class T>
inline std::vector<T> powers(T base, size_t n) {
    std::vector<T> out(n);
    constexpr size_t BLOCK = size_t(1) << 12;
    if (n < 16 * BLOCK) {
        T cur = T::one();
        for (size_t i = 0; i < n; i++) {
            out[i] = cur;
            cur = cur * base;
        }
        return out;
    }
    // long tables (the twiddles of a 2^25-point transform): the first entry of every block serially, the blocks in
    // parallel -- exact arithmetic, so the entries are the same field elements
    T step = T::one();
    for (size_t i = 0; i < BLOCK; i++) step = step * base;
    const size_t n_blocks = (n + BLOCK - 1) / BLOCK;
    std::vector<T> first(n_blocks);
    T cur = T::one();
    for (size_t b = 0; b < n_blocks; b++) {
        first[b] = cur;
        cur = cur * step;
    }
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < n_blocks; b++) {
        T c = first[b];
        const size_t hi = std::min(n, (b + 1) * BLOCK);
        for (size_t i = b * BLOCK; i < hi; i++) {
            out[i] = c;
            c = c * base;
        }
    }
    return out;
}

// Montgomery batch inversion is an optimisation in the reference
// (stark-backend/src/utils/batch_inverse.rs); results are plain inverses.
template <class T>
inline std::vector<T> batch_inverse(const std::vector<T> &xs) {
    std::vector<T> out(xs.size());
    for (size_t i = 0; i < xs.size(); i++) out[i] = xs[i].inv();
    return out;
}

inline unsigned log2_strict(size_t n) {
    assert(n && (n & (n - 1)) == 0);
    unsigned l = 0;
    while ((size_t(1) << l) < n) l++;
    return l;
}
inline unsigned log2_ceil(size_t n) {
    unsigned l = 0;
    while ((size_t(1) << l) < n) l++;
    return l;
}

}  // namespace oracle
