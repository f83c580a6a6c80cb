// ORACLE — test infrastructure only (see oracle/README.md).
//
// Polynomial helpers restated from the column-major SPEC implementation:
//   crates/stark-backend/src/prover/poly.rs:98-131    Mle::{evals_to_coeffs,coeffs_to_evals}_inplace
//   crates/stark-backend/src/prover/poly.rs:141-196   evals_eq_hypercube / mobius / hypercubes
//   crates/stark-backend/src/prover/poly.rs:325-348   eval_to_coeff_rs_message
//   crates/stark-backend/src/prover/poly.rs:371-457   UnivariatePoly::lagrange_interpolate
//   crates/stark-backend/src/prover/poly.rs:621-760   from_geometric_cosets_evals_idft
//   crates/stark-backend/src/poly_common.rs:6-232     eq / rot kernels, horner
// DFTs: the reference calls p3-dft (Radix2Bowers / Radix2DitParallel, =0.4.3, absent from the
// tree); only their mathematical result (natural order in, natural order out over <omega_n>)
// matters, so the oracle uses a textbook iterative radix-2.
#pragma once
#include <algorithm>
#include <omp.h>

#include "field.hpp"

namespace oracle {

template <class T>
inline void bit_reverse_inplace(std::vector<T> &a) {
    size_t n = a.size();
    unsigned lg = log2_strict(n);
    // (a LARGE transform called from serial code spreads over the cores itself: the BASELINE-size commitments are 25 columns
    // of 2^25 points on a 256-core host, and one thread per column left 90 % of the machine idle. Same arithmetic, same order
    // within every butterfly: results are bit-identical.)
    // (an `if` CLAUSE on the pragma still enters the OpenMP runtime for a team of one -- a fraction of a microsecond that the
    // 16-point transforms of eval_to_coeff pay a hundred million times at config D: 130 s of oracle time became 460 s on the
    // 256-core host. Small sizes take a plain loop, no pragma in sight.)
    auto body = [&](size_t i) {
        size_t j = 0;
        for (unsigned b = 0; b < lg; b++)
            if (i >> b & 1) j |= size_t(1) << (lg - 1 - b);
        if (i < j) std::swap(a[i], a[j]);  // (each unordered pair {i, j} is touched by the iteration i < j only)
    };
    if (n < (size_t(1) << 16) || omp_in_parallel()) {
        for (size_t i = 0; i < n; i++) body(i);
        return;
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) body(i);
}

// out[k] = sum_i a[i] * w^{ik}, w = two_adic_generator(log n) (or its inverse), natural order.
template <class T>
inline std::vector<T> dft_generic(std::vector<T> a, bool inverse) {
    size_t n = a.size();
    if (n <= 1) return a;
    unsigned lg = log2_strict(n);
    Fp w = two_adic_generator(lg);
    if (inverse) w = w.inv();
    bit_reverse_inplace(a);
    const bool par = n >= (size_t(1) << 16) && !omp_in_parallel();
    for (unsigned s = 1; s <= lg; s++) {
        size_t m = size_t(1) << s, h = m >> 1;
        Fp wm = w.exp_power_of_2(lg - s);
        std::vector<Fp> tw = powers(wm, h);
        if (!par) {
            for (size_t k = 0; k < n; k += m)
                for (size_t j = 0; j < h; j++) {
                    T t = a[k + j + h] * tw[j];
                    T u = a[k + j];
                    a[k + j] = u + t;
                    a[k + j + h] = u - t;
                }
            continue;
        }
        // n / 2 independent butterflies per stage: butterfly b = (block b / h, offset b % h)
#pragma omp parallel for schedule(static)
        for (size_t b = 0; b < n / 2; b++) {
            const size_t k = (b / h) * m, j = b % h;
            T t = a[k + j + h] * tw[j];
            T u = a[k + j];
            a[k + j] = u + t;
            a[k + j + h] = u - t;
        }
    }
    if (inverse) {
        Fp ninv = Fp((uint64_t)n).inv();
        if (!par) {
            for (auto &x : a) x = x * ninv;
        } else {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) a[i] = a[i] * ninv;
        }
    }
    return a;
}
template <class T>
inline std::vector<T> dft(std::vector<T> a) {
    return dft_generic(std::move(a), false);
}
template <class T>
inline std::vector<T> idft(std::vector<T> a) {
    return dft_generic(std::move(a), true);
}
// evaluations on shift * <omega_n>
template <class T>
inline std::vector<T> coset_dft(std::vector<T> a, Fp shift) {
    Fp s = Fp::one();
    for (auto &x : a) {
        x = x * s;
        s *= shift;
    }
    return dft(std::move(a));
}

// poly.rs:98-115
template <class T>
inline void mle_evals_to_coeffs_inplace(T *a, size_t len) {
    unsigned n = log2_strict(len);
    for (unsigned ls = 0; ls < n; ls++) {
        size_t step = size_t(1) << ls, span = step << 1;
        if (len < (size_t(1) << 16) || omp_in_parallel()) {
            for (size_t i = 0; i < len; i += span)
                for (size_t j = 0; j < step; j++) a[i + j + step] = a[i + j + step] - a[i + j];
            continue;
        }
        // len / 2 independent updates per level: update u = (block u / step, offset u % step)
#pragma omp parallel for schedule(static)
        for (size_t u = 0; u < len / 2; u++) {
            const size_t i = (u / step) * span, j = u % step;
            a[i + j + step] = a[i + j + step] - a[i + j];
        }
    }
}
// poly.rs:117-131
template <class T>
inline void mle_coeffs_to_evals_inplace(T *a, size_t len) {
    unsigned n = log2_strict(len);
    for (unsigned b = 0; b < n; b++) {
        size_t step = size_t(1) << b, span = step << 1;
        if (len < (size_t(1) << 16) || omp_in_parallel()) {
            for (size_t i = 0; i < len; i += span)
                for (size_t j = 0; j < step; j++) a[i + j + step] = a[i + j + step] + a[i + j];
            continue;
        }
        // len / 2 independent updates per level: update u = (block u / step, offset u % step)
#pragma omp parallel for schedule(static)
        for (size_t u = 0; u < len / 2; u++) {
            const size_t i = (u / step) * span, j = u % step;
            a[i + j + step] = a[i + j + step] + a[i + j];
        }
    }
}

// poly.rs:58-74 Mle::eval_at_point (coefficient form, little-endian monomials)
template <class T>
inline EF mle_coeffs_eval_at_point(const std::vector<T> &coeffs, const std::vector<EF> &x) {
    assert(coeffs.size() == size_t(1) << x.size());
    EF res = EF::zero();
    for (size_t i = 0; i < coeffs.size(); i++) {
        EF term = EF::one() * coeffs[i];
        for (size_t j = 0; j < x.size(); j++)
            if (i >> j & 1) term = term * x[j];
        res += term;
    }
    return res;
}

// poly.rs:141-155
template <class T>
inline std::vector<T> evals_eq_hypercube(const std::vector<T> &x) {
    size_t n = x.size();
    std::vector<T> out(size_t(1) << n);
    out[0] = T::one();
    for (size_t i = 0; i < n; i++) {
        const size_t half = size_t(1) << i;
        if (half < 4096 || omp_in_parallel()) {
            for (size_t y = 0; y < half; y++) {
                T lo = out[y];
                out[y + half] = lo * x[i];
                out[y] = lo * (T::one() - x[i]);
            }
            continue;
        }
#pragma omp parallel for schedule(static)
        for (size_t y = 0; y < half; y++) {
            T lo = out[y];
            out[y + half] = lo * x[i];
            out[y] = lo * (T::one() - x[i]);
        }
    }
    return out;
}

// poly.rs:165-182 ; K_i(0) = 1 - 2u_i, K_i(1) = u_i
inline std::vector<EF> evals_mobius_eq_hypercube(const std::vector<EF> &u) {
    size_t n = u.size();
    std::vector<EF> out(size_t(1) << n);
    out[0] = EF::one();
    for (size_t i = 0; i < n; i++) {
        EF w0 = EF::one() - u[i].dbl(), w1 = u[i];
        if ((size_t(1) << i) < 4096 || omp_in_parallel()) {
            for (size_t y = 0; y < (size_t(1) << i); y++) {
                EF prev = out[y];
                out[y + (size_t(1) << i)] = prev * w1;
                out[y] = prev * w0;
            }
            continue;
        }
#pragma omp parallel for schedule(static)
        for (size_t y = 0; y < (size_t(1) << i); y++) {
            EF prev = out[y];
            out[y + (size_t(1) << i)] = prev * w1;
            out[y] = prev * w0;
        }
    }
    return out;
}

// poly.rs:186-196 : concatenated segment tree, layer i at offset 2^i - 1, reversed endianness
inline std::vector<EF> evals_eq_hypercubes(size_t n, const std::vector<EF> &x /* n entries */) {
    std::vector<EF> out((size_t(2) << n) - 1);
    out[0] = EF::one();
    for (size_t i = 0; i < x.size(); i++)
        for (size_t y = 0; y < (size_t(1) << i); y++) {
            EF prev = out[(size_t(1) << i) - 1 + y];
            out[(size_t(1) << (i + 1)) - 1 + (2 * y + 1)] = prev * x[i];
            out[(size_t(1) << (i + 1)) - 1 + (2 * y)] = prev * (EF::one() - x[i]);
        }
    return out;
}

// poly_common.rs:6-20
template <class A, class B>
inline EF eval_eq_mle(const std::vector<A> &x, const std::vector<B> &y) {
    assert(x.size() == y.size());
    EF acc = EF::one();
    for (size_t i = 0; i < x.size(); i++) {
        EF xi = EF::one() * x[i], yi = EF::one() * y[i];
        acc = acc * (EF::one() - yi - xi + (xi * yi).dbl());
    }
    return acc;
}

// poly_common.rs:56-62
template <class T>
inline T eval_eq_uni(unsigned l_skip, T x, T y) {
    T res = T::one();
    T xp = x, yp = y;
    for (unsigned i = 0; i < l_skip; i++) {
        res = (xp + yp) * res + (xp - T::one()) * (yp - T::one());
        xp = xp.square();
        yp = yp.square();
    }
    return res * Fp::one().halve().pow(l_skip);
}
// poly_common.rs:67-73
template <class T>
inline T eval_eq_uni_at_one(unsigned l_skip, T x) {
    T res = T::one();
    T xp = x;
    for (unsigned i = 0; i < l_skip; i++) {
        res = res * (xp + T::one());
        xp = xp.square();
    }
    return res * Fp::one().halve().pow(l_skip);
}
// poly_common.rs:100-110
template <class T>
inline T eval_in_uni(unsigned l_skip, int n, T z) {
    if (n < 0) return eval_eq_uni_at_one((unsigned)(-n), z.exp_power_of_2(l_skip + n));
    return T::one();
}
// poly_common.rs:82-98 : eq_D(x, Z) as coefficients in Z
inline std::vector<EF> eq_uni_poly(unsigned l_skip, EF x) {
    Fp n_inv = Fp::one().halve().pow(l_skip);
    size_t N = size_t(1) << l_skip;
    std::vector<EF> coeffs(N);
    EF xp = x;
    for (size_t i = 0; i < N; i++) {
        coeffs[i] = xp * n_inv;
        xp = xp * x;
    }
    std::reverse(coeffs.begin(), coeffs.end());
    coeffs[0] = EF(n_inv);
    return coeffs;
}
// poly_common.rs:131-164
inline EF eval_eq_sharp_uni(const std::vector<Fp> &omega_skip_pows, const std::vector<EF> &xi_1,
                            EF z) {
    unsigned l_skip = (unsigned)xi_1.size();
    std::vector<EF> eq_xi = evals_eq_hypercube(xi_1);
    EF res = EF::zero();
    for (size_t i = 0; i < omega_skip_pows.size(); i++)
        res += eval_eq_uni(l_skip, z, EF(omega_skip_pows[i])) * eq_xi[i];
    return res;
}
// poly.rs:226-229
inline std::vector<EF> eq_sharp_uni_poly(const std::vector<EF> &xi_1) {
    return idft(evals_eq_hypercube(xi_1));
}

// poly_common.rs:184-196 : MLE of (eq, cyclic-rotation kernel) on the hypercube
inline std::pair<EF, EF> eval_eq_rot_cube(const std::vector<EF> &x, const std::vector<EF> &y) {
    EF rot = EF::one(), eq = EF::one();
    for (size_t k = x.size(); k-- > 0;) {
        rot = x[k] * (EF::one() - y[k]) * eq + (EF::one() - x[k]) * y[k] * rot;
        eq = eq * (x[k] * y[k] + (EF::one() - x[k]) * (EF::one() - y[k]));
    }
    return {eq, rot};
}
// poly_common.rs:169-181
inline EF eval_rot_kernel_prism(unsigned l_skip, const std::vector<EF> &x, const std::vector<EF> &y) {
    EF omega = EF(two_adic_generator(l_skip));
    std::vector<EF> xr(x.begin() + 1, x.end()), yr(y.begin() + 1, y.end());
    auto [eq_cube, rot_cube] = eval_eq_rot_cube(xr, yr);
    return eval_eq_uni(l_skip, x[0], y[0] * omega) * eq_cube +
           eval_eq_uni_at_one(l_skip, x[0]) * eval_eq_uni_at_one(l_skip, y[0] * omega) *
               (rot_cube - eq_cube);
}

template <class C, class X>
inline EF horner_eval(const std::vector<C> &coeffs, X x) {
    EF acc = EF::zero();
    for (size_t i = coeffs.size(); i-- > 0;) acc = acc * x + coeffs[i];
    return acc;
}

// poly.rs:325-348 : per 2^l_skip chunk, iDFT then subset-zeta over the chunk bits only
inline std::vector<Fp> eval_to_coeff_rs_message(unsigned l_skip, const Fp *evals, size_t len) {
    size_t chunk = size_t(1) << l_skip;
    assert(len % chunk == 0);
    std::vector<Fp> buf(len);
    // (the chunks of ONE tall column over the cores when the caller is serial; inside, every transform is tiny and serial)
#pragma omp parallel for schedule(static) if (len >= (size_t(1) << 20) && !omp_in_parallel())
    for (size_t off = 0; off < len; off += chunk) {
        std::vector<Fp> c(evals + off, evals + off + chunk);
        c = idft(std::move(c));
        mle_coeffs_to_evals_inplace(c.data(), chunk);
        std::copy(c.begin(), c.end(), buf.begin() + off);
    }
    return buf;
}

// poly.rs:371-457 : coefficient form through (points[i], evals[i])
inline std::vector<EF> lagrange_interpolate(const std::vector<Fp> &points,
                                            const std::vector<EF> &evals) {
    size_t len = points.size();
    assert(evals.size() == len);
    std::vector<EF> coeffs(len, EF::zero());
    if (len == 0) return coeffs;
    if (len == 1) return {evals[0]};
    for (size_t i = 0; i < len; i++) {
        // L_i(x) = prod_{j != i} (x - p_j) / (p_i - p_j)
        std::vector<Fp> l{Fp::one()};
        for (size_t j = 0; j < len; j++) {
            if (j == i) continue;
            Fp scale = (points[i] - points[j]).inv();
            l.push_back(Fp::zero());
            for (size_t k = l.size() - 1; k >= 1; k--) {
                Fp prev = l[k - 1] * scale;
                l[k] += prev;
                l[k - 1] = -(prev * points[j]);
            }
        }
        for (size_t k = 0; k < len; k++) coeffs[k] += evals[i] * l[k];
    }
    return coeffs;
}

// poly.rs:621-683 : `evals` is row-major height x width (row = point index in D, col = coset
// index); coset i is init * shift^i * D. Returns coefficients of the unique poly of degree
// < height*width. Done here by direct Lagrange interpolation across cosets for each residue
// class of the exponent (same maths as the reference; no basis caching).
inline std::vector<EF> from_geometric_cosets_evals_idft(const std::vector<EF> &evals, size_t height,
                                                        size_t width, Fp shift, Fp init) {
    if (height == 0 || width == 0) return {};
    unsigned log_h = log2_strict(height);
    // per coset: interpolate remainder mod X^h - (init*shift^i)^h, then unshift coefficients
    std::vector<std::vector<EF>> rem(width);
    for (size_t c = 0; c < width; c++) {
        std::vector<EF> col(height);
        for (size_t r = 0; r < height; r++) col[r] = evals[r * width + c];
        col = idft(std::move(col));
        Fp s_inv = (init * shift.pow(c)).inv();
        Fp sp = Fp::one();
        for (size_t r = 0; r < height; r++) {
            col[r] = col[r] * sp;
            sp *= s_inv;
        }
        rem[c] = std::move(col);
    }
    // rem[c][t] = sum_k coeff[k*height + t] * (y_c)^k with y_c = (init*shift^c)^height
    std::vector<Fp> ys(width);
    for (size_t c = 0; c < width; c++) ys[c] = (init * shift.pow(c)).exp_power_of_2(log_h);
    std::vector<EF> coeffs(height * width);
    for (size_t t = 0; t < height; t++) {
        std::vector<EF> vals(width);
        for (size_t c = 0; c < width; c++) vals[c] = rem[c][t];
        std::vector<EF> pc = lagrange_interpolate(ys, vals);
        for (size_t k = 0; k < width; k++) coeffs[k * height + t] = pc[k];
    }
    return coeffs;
}

}  // namespace oracle
