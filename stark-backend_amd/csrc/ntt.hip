// BabyBear NTT for gfx950: eval->coeff message transform and Reed-Solomon (coset LDE) encode.
//
// Replaces (reference): crates/cuda-backend/cuda/supra/ntt.cu:22-254 (mixed-radix CT NTT),
// supra/ntt_bitrev.cu (bit reversal), src/batch_ntt_small.cu:78-114 (chunk iNTT),
// src/mle_interpolate.cu (subset-zeta), src/matrix.cu:187-212 (zero-extension) as driven by
// crates/cuda-backend/src/stacked_pcs.rs:229-337 (`rs_code_matrix`).
//
// Layout: matrices are column-major planes, row index fastest: a wavefront's 64 lanes touch 64
// consecutive rows of one column = 256 contiguous bytes. The codeword is produced by a
// decimation-in-frequency transform and is LEFT in bit-reversed row order ("physical order"):
//   physical row p  <->  natural evaluation index m = bitrev_{L}(p),  L = log2(rows).
// That removes the bit-reversal pass altogether, and it is also the order the Merkle tree wants:
// the 2^k rows opened by one WHIR query {i + t*stride} are 2^k CONSECUTIVE physical rows.
//
// A pass handles `r` consecutive DIF stages on tiles of 2^r x 2^c elements (2^c consecutive rows
// so that every global access is a run of >= 2^c * 4 bytes) staged through LDS. Zero padding of
// the message is virtual: rows >= 2^log_n_in read as zero in the first pass.
#include <mutex>

#include "device_ctx.h"
#include "openvm_hip.h"
#include "tuning.h"
#include "wave_chunks.cuh"

using bb::Fp;
using bb::FpExt;

namespace ovm {

static constexpr int NTT_THREADS = 256;
static constexpr int NTT_TILE_LOG = 13;  // 8192 base-field elements (32 KiB of LDS) per workgroup

namespace {
struct TwState {
    Twiddles dev;
    bool ready = false;
};
TwState g_tw[16];
}  // namespace

const Twiddles *twiddles() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    static std::mutex mu;  // engines on several host threads may initialise the same device at once
    std::lock_guard<std::mutex> lk(mu);
    TwState &st = g_tw[dev];
    if (st.ready) return &st.dev;
    const size_t nlo = size_t(1) << TW_LO_BITS, nhi = size_t(1) << TW_HI_BITS;
    Fp *h = (Fp *)malloc((nlo + nhi) * sizeof(Fp));
    if (!h) return nullptr;
    Fp w = bb::two_adic_generator(TW_LOG_MAX);
    Fp cur = Fp::one();
    for (size_t i = 0; i < nlo; i++) {
        h[i] = cur;
        cur *= w;
    }
    Fp wh = cur;  // w^(2^12)
    cur = Fp::one();
    for (size_t i = 0; i < nhi; i++) {
        h[nlo + i] = cur;
        cur *= wh;
    }
    Fp *d = nullptr;
    if (hipMalloc(&d, (nlo + nhi) * sizeof(Fp)) != hipSuccess) {
        free(h);
        return nullptr;
    }
    hipError_t e = hipMemcpy(d, h, (nlo + nhi) * sizeof(Fp), hipMemcpyHostToDevice);
    free(h);
    if (e != hipSuccess) return nullptr;
    st.dev.lo = d;
    st.dev.hi = d + nlo;
    st.ready = true;
    return &st.dev;
}

// ---------------------------------------------------------------------------------------------
// Message transform: for every 2^l_skip chunk of every column, inverse DFT over the chunk
// (natural order coefficients) followed by the subset-zeta transform over the chunk bits
// (reference algorithm: stark-backend/src/prover/poly.rs:325-348).
// A workgroup stages 2^tile_log contiguous rows of one column in LDS. The inverse transform is
// DIF with inverse twiddles (natural in, bit-reversed out); the zeta transform is invariant under
// the bit permutation, so it runs on the bit-reversed data and the permutation is undone on the
// way out.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTT_THREADS) void eval_to_coeff_kernel(
    const Fp *__restrict__ in, Fp *__restrict__ out, uint32_t log_height, uint32_t l_skip,
    uint32_t tile_log, Fp inv_chunk, Twiddles tw) {
    extern __shared__ Fp tile[];
    const uint32_t tile_size = 1u << tile_log;
    const uint64_t col_off = (uint64_t)blockIdx.y << log_height;
    const uint64_t base = col_off + ((uint64_t)blockIdx.x << tile_log);
    for (uint32_t i = threadIdx.x; i < tile_size; i += NTT_THREADS) tile[i] = in[base + i];
    __syncthreads();
    const uint32_t half_count = tile_size >> 1;
    // inverse DIF over the low l_skip bits
    for (uint32_t k = 0; k < l_skip; k++) {
        const uint32_t bit = l_skip - 1 - k;  // paired bit
        for (uint32_t bf = threadIdx.x; bf < half_count; bf += NTT_THREADS) {
            uint32_t lo_mask = (1u << bit) - 1;
            uint32_t i0 = ((bf & ~lo_mask) << 1) | (bf & lo_mask);
            uint32_t i1 = i0 | (1u << bit);
            // twiddle: omega_{2^(bit+1)}^(-(i0 mod 2^bit))
            uint32_t e = i0 & lo_mask;
            uint32_t e27 = (uint32_t)(((1u << TW_LOG_MAX) - ((uint64_t)e << (TW_LOG_MAX - bit - 1))) &
                                      ((1u << TW_LOG_MAX) - 1));
            Fp a = tile[i0], b = tile[i1];
            tile[i0] = a + b;
            Fp d = a - b;
            tile[i1] = e ? d * tw_pow27(tw, e27) : d;
        }
        __syncthreads();
    }
    // subset-zeta over the chunk bits (physical bit b == logical bit l_skip-1-b; order-free)
    for (uint32_t bit = 0; bit < l_skip; bit++) {
        for (uint32_t bf = threadIdx.x; bf < half_count; bf += NTT_THREADS) {
            uint32_t lo_mask = (1u << bit) - 1;
            uint32_t i0 = ((bf & ~lo_mask) << 1) | (bf & lo_mask);
            uint32_t i1 = i0 | (1u << bit);
            tile[i1] = tile[i1] + tile[i0];
        }
        __syncthreads();
    }
    const uint32_t chunk_mask = (1u << l_skip) - 1;
    for (uint32_t i = threadIdx.x; i < tile_size; i += NTT_THREADS) {
        uint32_t src = (i & ~chunk_mask) | bitrev32(i & chunk_mask, l_skip);
        out[base + i] = tile[src] * inv_chunk;
    }
}

// Register form of the message transform for 16-row chunks (l_skip = 4): a lane owns one chunk,
// all twiddles are compile-time powers of omega_16^-1, no LDS. In register i sits the coefficient
// of degree bitrev4(i) after the DIF; the zeta transform runs on register-index bits (it is
// invariant under the bit permutation) and the store undoes the permutation.
#define W16P(j) bb::cx_to_mont(bb::cx_powmod(bb::TWO_ADIC_GENERATORS[4], (j)))
static constexpr uint32_t W16_INV[8] = {W16P(0), W16P(15), W16P(14), W16P(13), W16P(12), W16P(11), W16P(10), W16P(9)};
#undef W16P
// XPOSE (a whole number of workgroups of chunks): the chunks change lanes in LDS on the way in and out, so that a wave's
// loads and stores are whole 1 KiB runs (wave_chunks.cuh).
template <bool XPOSE>
__global__ __launch_bounds__(256) void eval_to_coeff16_kernel(const Fp *__restrict__ in, Fp *__restrict__ out,
                                                              uint64_t n_chunks, Fp inv_chunk) {
    __shared__ uint4 tile[XPOSE ? 4 * WAVE_CHUNK_TILE : 1];
    const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= n_chunks) return;
    const uint32_t lane = threadIdx.x & 63;
    uint4 *T = tile + (XPOSE ? (threadIdx.x >> 6) * WAVE_CHUNK_TILE : 0);
    Fp x[16];
    const uint4 *src = reinterpret_cast<const uint4 *>(in + (chunk << 4));
    uint4 own[4];
    if (XPOSE) {
        wave_chunks_load(T, reinterpret_cast<const uint4 *>(in + ((chunk - lane) << 4)), lane, own);
    } else {
#pragma unroll
        for (int v = 0; v < 4; v++) own[v] = src[v];
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
        uint4 q = own[v];
        x[4 * v] = Fp::raw(q.x);
        x[4 * v + 1] = Fp::raw(q.y);
        x[4 * v + 2] = Fp::raw(q.z);
        x[4 * v + 3] = Fp::raw(q.w);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int bit = 3 - k, half = 1 << bit;
#pragma unroll
        for (int j = 0; j < half; j++) {
#pragma unroll
            for (int g = 0; g < 16; g += 2 * half) {
                Fp a = x[g + j], b = x[g + j + half];
                x[g + j] = a + b;
                Fp d = a - b;
                x[g + j + half] = j == 0 ? d : d * Fp::raw(W16_INV[j << (3 - bit)]);
            }
        }
    }
#pragma unroll
    for (int bit = 0; bit < 4; bit++) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (i & (1 << bit)) x[i] = x[i] + x[i ^ (1 << bit)];
    }
    auto br = [](int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); };
    uint4 *dst = reinterpret_cast<uint4 *>(out + (chunk << 4));
#pragma unroll
    for (int v = 0; v < 4; v++)
        own[v] = make_uint4((x[br(4 * v)] * inv_chunk).v, (x[br(4 * v + 1)] * inv_chunk).v,
                            (x[br(4 * v + 2)] * inv_chunk).v, (x[br(4 * v + 3)] * inv_chunk).v);
    if (XPOSE) {
        wave_chunks_store(T, reinterpret_cast<uint4 *>(out + ((chunk - lane) << 4)), lane, own);
    } else {
#pragma unroll
        for (int v = 0; v < 4; v++) dst[v] = own[v];
    }
}

// ---------------------------------------------------------------------------------------------
// One DIF pass: stages s0 .. s0+r-1 of a 2^L-point transform on every column.
//   position p = (hi << (beta + r)) | (t << beta) | (mid << c) | cc ,  beta = L - s0 - r >= c
// Stage s pairs positions differing in bit L-1-s; the lower one (bit clear) at position p uses
// twiddle omega_{2^(L-s)}^(p mod 2^(L-1-s)).
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(NTT_THREADS) void dif_pass_kernel(const T *in, T *out, uint32_t L,
                                                               uint32_t log_n_in, uint32_t s0,
                                                               uint32_t r, uint32_t c,
                                                               uint64_t in_col_stride,
                                                               uint64_t out_col_stride,
                                                               Twiddles tw) {
    extern __shared__ unsigned char smem_raw[];
    T *tile = reinterpret_cast<T *>(smem_raw);
    const uint32_t beta = L - s0 - r;
    const uint32_t tile_size = 1u << (r + c);
    const uint32_t c_mask = (1u << c) - 1;
    const uint64_t tile_id = blockIdx.x;
    const uint64_t mid = tile_id & ((1ull << (beta - c)) - 1);
    const uint64_t hi = tile_id >> (beta - c);
    const uint64_t pos_base = (hi << (beta + r)) | (mid << c);
    const T *in_col = in + (uint64_t)blockIdx.y * in_col_stride;
    T *out_col = out + (uint64_t)blockIdx.y * out_col_stride;
    const uint64_t n_in = 1ull << log_n_in;

    for (uint32_t l = threadIdx.x; l < tile_size; l += NTT_THREADS) {
        uint32_t cc = l & c_mask, t = l >> c;
        uint64_t p = pos_base | ((uint64_t)t << beta) | cc;
        T v;
        if (p < n_in)
            v = in_col[p];
        else
            v = T::zero();
        tile[l] = v;
    }
    __syncthreads();

    const uint32_t half_count = tile_size >> 1;
    for (uint32_t k = 0; k < r; k++) {
        const uint32_t s = s0 + k;
        const uint32_t tbit = r - 1 - k;  // paired bit of t
        const uint32_t lbit = tbit + c;   // paired bit of the local index
        const uint32_t lo_mask = (1u << lbit) - 1;
        const uint32_t shift = TW_LOG_MAX - (L - s);  // omega_{2^(L-s)} = omega_27^(1 << shift)
        for (uint32_t bf = threadIdx.x; bf < half_count; bf += NTT_THREADS) {
            uint32_t l0 = ((bf & ~lo_mask) << 1) | (bf & lo_mask);
            uint32_t l1 = l0 | (1u << lbit);
            uint32_t cc = l0 & c_mask, t = l0 >> c;
            // e = p mod 2^(L-1-s) for the lower element
            uint64_t p = pos_base | ((uint64_t)t << beta) | cc;
            uint32_t e = (uint32_t)(p & ((1ull << (L - 1 - s)) - 1));
            T a = tile[l0], b = tile[l1];
            tile[l0] = a + b;
            T d = a - b;
            tile[l1] = e ? d * tw_pow27(tw, e << shift) : d;
        }
        __syncthreads();
    }

    for (uint32_t l = threadIdx.x; l < tile_size; l += NTT_THREADS) {
        uint32_t cc = l & c_mask, t = l >> c;
        uint64_t p = pos_base | ((uint64_t)t << beta) | cc;
        out_col[p] = tile[l];
    }
}

// ---------------------------------------------------------------------------------------------
// Radix-16 register passes (transforms of 2^9 points and more).
// A lane owns 16 elements that differ in a 4-bit window of the tile index and runs 4 DIF stages
// on them in registers; LDS is touched once per 4 stages (exchange), not once per stage.
// Twiddles of the window whose lowest bit is global position bit b (S = 2^b):
//   stage k (k = 0..3) uses  w^(2^k) * omega_(16 >> k)^j ,  w = omega_(16 S)^(p mod S),
// i.e. ONE table lookup per lane per window; the small roots are compile-time constants. In the
// last window of the contiguous pass S = 1, so it needs no data-dependent twiddle at all.
// ---------------------------------------------------------------------------------------------
#define W16C(j) bb::cx_to_mont(bb::cx_powmod(bb::TWO_ADIC_GENERATORS[4], j))
static constexpr uint32_t W16[8] = {W16C(0), W16C(1), W16C(2), W16C(3), W16C(4), W16C(5), W16C(6), W16C(7)};
#undef W16C

// (a - b) * w in SIGNED Montgomery form (bb31.h: smont_mul -- 5 VALU, no correction step): the difference is taken as the
// signed representative a - b in (-P, P) (one subtract), the twiddle may itself be a signed representative, and one
// add + min canonicalises the product: 8 VALU per twiddled butterfly output instead of 9, and the twiddle products of a
// window (14 per lane: w^(2^k) * omega_16^j) stay signed at 5 VALU instead of 7 -- 12 % fewer instructions per radix-16
// window of a kernel that is VALU-bound.
struct STw {  // a twiddle as a signed representative in (-P, P)
    int32_t v;
};
__device__ __forceinline__ Fp mul_diff(Fp a, Fp b, STw w) {
    return Fp::raw(bb::from_signed(bb::smont_mul((int32_t)(a.v - b.v), w.v)));
}
__device__ __forceinline__ FpExt mul_diff(const FpExt &a, const FpExt &b, STw w) {
    FpExt r;
#pragma unroll
    for (int k = 0; k < 4; k++) r.c[k] = mul_diff(a.c[k], b.c[k], w);
    return r;
}

// NB stages on the low NB bits of the register index (upper bits = independent groups).
template <class T, int NB, bool HAS_W>
__device__ __forceinline__ void radix_dif(T (&x)[16], Fp w0) {
    STw wk{(int32_t)w0.v};
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int bit = NB - 1 - k, half = 1 << bit;
#pragma unroll
        for (int j = 0; j < half; j++) {
            const int32_t cj = (int32_t)W16[j << (3 - bit)];
            const STw tw = HAS_W ? (j == 0 ? wk : STw{bb::smont_mul(wk.v, cj)}) : STw{cj};
#pragma unroll
            for (int g = 0; g < 16; g += 2 * half) {
                const int v = g + j;
                T a = x[v], b = x[v + half];
                x[v] = a + b;
                x[v + half] = (!HAS_W && j == 0) ? a - b : mul_diff(a, b, tw);
            }
        }
        if (HAS_W) wk = STw{bb::smont_mul(wk.v, wk.v)};
    }
}

// One pass over tiles of 2^TL elements. Strided form (c > 0): r = 8 stages s0..s0+7, tile =
// 2^8 x 2^c. Contiguous form (c == 0): the last `r` stages of the transform on 2^TL consecutive
// positions, as windows of 4 from the top and a final window [3..0] doing the r mod 4 (or 4)
// lowest stages.
template <class T, int TL>
__global__ __launch_bounds__(1 << (TL - 4)) void dif16_pass_kernel(const T *in, T *out, uint32_t L,
                                                                   uint32_t log_n_in, uint32_t s0, uint32_t r,
                                                                   uint32_t c, uint64_t in_col_stride,
                                                                   uint64_t out_col_stride, Twiddles tw) {
    // Padded by one element per 16 so that a lane reading its 16 consecutive elements (last window)
    // and lanes reading consecutive elements (all other windows) are both bank-conflict free.
    __shared__ T tile[(1 << TL) + (1 << (TL - 4))];
    auto pad = [](uint32_t l) -> uint32_t { return l + (l >> 4); };
    const uint32_t tid = threadIdx.x;
    const uint32_t beta = L - s0 - r;  // global bit of the lowest stage of this pass
    uint64_t pos_base;
    if (c > 0) {
        const uint64_t mid = blockIdx.x & ((1ull << (beta - c)) - 1), hi = blockIdx.x >> (beta - c);
        pos_base = (hi << (beta + r)) | (mid << c);
    } else {
        pos_base = (uint64_t)blockIdx.x << TL;
    }
    const uint32_t c_mask = (1u << c) - 1;
    auto pos_of = [&](uint32_t l) -> uint64_t {
        return c > 0 ? (pos_base | ((uint64_t)(l >> c) << beta) | (l & c_mask)) : (pos_base | l);
    };
    auto gbit_of = [&](uint32_t lbit) -> uint32_t { return (c > 0 && lbit >= c) ? beta + (lbit - c) : lbit; };
    const T *in_col = in + (uint64_t)blockIdx.y * in_col_stride;
    T *out_col = out + (uint64_t)blockIdx.y * out_col_stride;
    const uint64_t n_in = 1ull << log_n_in;

    // window list: full windows from the top of the pass's stage bits, then the final one
    const uint32_t top = c > 0 ? c + r : r;          // one past the highest active tile bit
    const uint32_t n_full = c > 0 ? r / 4 : (r - 1) / 4;  // contiguous: keep 1..4 stages for the last window
    const uint32_t last_nb = c > 0 ? 0 : r - 4 * n_full;  // 1..4 (contiguous only)
    T x[16];
    uint32_t bl = top - 4;
    auto l_of = [&](uint32_t v) -> uint32_t {
        return ((tid >> bl) << (bl + 4)) | (v << bl) | (tid & ((1u << bl) - 1));
    };
    // LDS index of register v: bits [bl, bl + 4) of l_of(0) are clear, so l_of(v) = l_of(0) + (v << bl) without a carry and
    // pad() splits: pad(l_of(v)) = pad(l_of(0)) + x + (x >> 4), x = v << bl -- a lane part computed once per window plus a
    // WAVE-UNIFORM part (scalar registers), i.e. one vector add per access instead of the shifts, ors and the add of pad()
    // on every one of the 32 accesses of a window (they were a third of the pass's vector instructions).
    auto lds_of = [&](uint32_t a0, uint32_t v) -> uint32_t {
        const uint32_t x = v << bl;
        return a0 + x + (x >> 4);
    };
    const uint32_t n_windows = n_full + (last_nb ? 1 : 0);
    for (uint32_t wdx = 0; wdx < n_windows; wdx++) {
        const bool is_last_partial = (last_nb != 0) && (wdx == n_full);
        bl = is_last_partial ? 0 : top - 4 * (wdx + 1);
        if (wdx == 0) {
            // Position of register v: a uniform part (pos_base + v * vstep, folded into a scalar base
            // address) + a lane part that fits 32 bits -- one VGPR offset for all 16 loads instead of
            // ~11 vector instructions of 64-bit index arithmetic (and a branch) per load.
            const uint32_t l0 = l_of(0);
            const uint32_t lane_off = c > 0 ? (uint32_t)((((uint64_t)(l0 >> c)) << beta) | (l0 & c_mask)) : l0;
            const uint64_t vstep = c > 0 ? 1ull << (bl - c + beta) : 1ull << bl;
            const T *src0 = in_col + pos_base;
            if (log_n_in >= L) {  // every position holds an input element
#pragma unroll
                for (int v = 0; v < 16; v++) x[v] = (src0 + (uint64_t)v * vstep)[lane_off];
            } else if (c > 0) {
                // zero-padded input, strided pass: window 0 is the top of the tile, so lane_off < vstep and
                // register v is in range for all lanes or for none when n_in is a multiple of vstep
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const uint64_t ub = pos_base + (uint64_t)v * vstep;
                    if (ub + vstep <= n_in) x[v] = (src0 + (uint64_t)v * vstep)[lane_off];
                    else if (ub >= n_in) x[v] = T::zero();
                    else x[v] = ub + lane_off < n_in ? (src0 + (uint64_t)v * vstep)[lane_off] : T::zero();
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    uint64_t p = pos_of(l_of(v));
                    x[v] = p < n_in ? in_col[p] : T::zero();
                }
            }
        } else {
            const uint32_t a0 = pad(l_of(0));
#pragma unroll
            for (int v = 0; v < 16; v++) x[v] = tile[lds_of(a0, v)];
            __syncthreads();
        }
        if (is_last_partial) {
            switch (last_nb) {
                case 1: radix_dif<T, 1, false>(x, Fp::one()); break;
                case 2: radix_dif<T, 2, false>(x, Fp::one()); break;
                case 3: radix_dif<T, 3, false>(x, Fp::one()); break;
                default: radix_dif<T, 4, false>(x, Fp::one()); break;
            }
        } else {
            const uint32_t gb = gbit_of(bl);  // S = 2^gb
            const uint64_t p0 = pos_of(l_of(0));
            const uint32_t q = (uint32_t)(p0 & ((1ull << gb) - 1));
            Fp w0 = tw_pow27(tw, q << (TW_LOG_MAX - gb - 4));
            radix_dif<T, 4, true>(x, w0);
        }
        if (wdx + 1 < n_windows) {
            const uint32_t a0 = pad(l_of(0));
#pragma unroll
            for (int v = 0; v < 16; v++) tile[lds_of(a0, v)] = x[v];
            __syncthreads();
        }
    }
    if (c == 0 && bl == 0) {
        // the last window left every lane with 16 CONSECUTIVE positions: transpose through LDS so
        // that consecutive lanes store consecutive positions (256 B per wave-store)
        const uint32_t a0 = pad(l_of(0));
#pragma unroll
        for (int v = 0; v < 16; v++) tile[lds_of(a0, v)] = x[v];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 16; v++) {
            uint32_t l = (uint32_t)v * (1u << (TL - 4)) + tid;
            out_col[pos_base | l] = tile[pad(l)];
        }
        return;
    }
    {  // (strided pass) same split as the loads: scalar base per register, one lane offset
        const uint32_t l0 = l_of(0);
        const uint32_t lane_off = c > 0 ? (uint32_t)((((uint64_t)(l0 >> c)) << beta) | (l0 & c_mask)) : l0;
        const uint64_t vstep = c > 0 && bl >= c ? 1ull << (bl - c + beta) : 0;
        if (vstep) {
            T *dst0 = out_col + pos_base;
#pragma unroll
            for (int v = 0; v < 16; v++) (dst0 + (uint64_t)v * vstep)[lane_off] = x[v];
            return;
        }
    }
#pragma unroll
    for (int v = 0; v < 16; v++) out_col[pos_of(l_of(v))] = x[v];
}

template <class T, int TL>
static int launch_dif16(const T *src, T *out, uint64_t width, uint32_t L, uint32_t log_n_in, uint32_t s0, uint32_t r,
                        uint32_t c, uint64_t src_stride, uint64_t out_stride, const Twiddles &tw, hipStream_t stream) {
    dim3 grid(1u << (L - TL), (unsigned)width);
    OVM_LAUNCH((dif16_pass_kernel<T, TL>), grid, dim3(1 << (TL - 4)), 0, stream, src, out, L, log_n_in, s0, r,
                       c, src_stride, out_stride, tw);
    OVM_CHECK_LAUNCH();
    return 0;
}

template <class T>
static int dif_ntt16(const T *in, T *out, uint64_t width, uint32_t L, uint32_t log_n_in, uint64_t in_col_stride,
                     uint64_t out_col_stride, const Twiddles &tw, hipStream_t stream) {
    constexpr bool BASE = sizeof(T) == 4;
    const uint32_t TL = BASE ? (L >= 13 ? 13 : (L >= 11 ? 11 : 9)) : (L >= 11 ? 11 : 9);
    const uint32_t ns = L > TL ? (L - TL + 7) / 8 : 0;  // strided passes of 8 stages
    const uint32_t rc = L - 8 * ns;                      // stages of the contiguous pass (1..TL)
    const uint32_t c = TL - 8;
    OVM_PROF(BASE ? "ntt.dif_passes" : "ntt.dif_passes_ext", stream);
    const T *src = in;
    uint64_t src_stride = in_col_stride;
    uint32_t src_log_n = log_n_in, s = 0;
    auto launch = [&](uint32_t r, uint32_t cc) -> int {
        if (BASE) {
            if (TL == 13) return launch_dif16<T, 13>(src, out, width, L, src_log_n, s, r, cc, src_stride, out_col_stride, tw, stream);
            if (TL == 11) return launch_dif16<T, 11>(src, out, width, L, src_log_n, s, r, cc, src_stride, out_col_stride, tw, stream);
            return launch_dif16<T, 9>(src, out, width, L, src_log_n, s, r, cc, src_stride, out_col_stride, tw, stream);
        }
        if (TL == 11) return launch_dif16<T, 11>(src, out, width, L, src_log_n, s, r, cc, src_stride, out_col_stride, tw, stream);
        return launch_dif16<T, 9>(src, out, width, L, src_log_n, s, r, cc, src_stride, out_col_stride, tw, stream);
    };
    for (uint32_t i = 0; i < ns; i++) {
        int rc_ = launch(8, c);
        if (rc_) return rc_;
        s += 8;
        src = out;
        src_stride = out_col_stride;
        src_log_n = L;
    }
    return launch(rc, 0);
}

// Plans and launches the passes of a 2^L-point DIF on `width` columns.
template <class T>
static int dif_ntt(const T *in, T *out, uint64_t width, uint32_t L, uint32_t log_n_in,
                   uint64_t in_col_stride, uint64_t out_col_stride, hipStream_t stream) {
    const Twiddles *tw = twiddles();
    if (!tw) return (int)hipErrorInvalidValue;
    if (L > TW_LOG_MAX) return (int)hipErrorInvalidValue;
    if (width == 0) return 0;
    if (L == 0) {
        // single point: copy (or zero)
        for (uint64_t j = 0; j < width; j++)
            OVM_CHECK(hipMemcpyAsync(out + j * out_col_stride, in + j * in_col_stride, sizeof(T),
                                     hipMemcpyDeviceToDevice, stream));
        return 0;
    }
    if (L >= 9) return dif_ntt16<T>(in, out, width, L, log_n_in, in_col_stride, out_col_stride, *tw, stream);
    // Elements per tile: 32 KiB of LDS for both Fp and FpExt tiles.
    const uint32_t tile_log = sizeof(T) == 4 ? NTT_TILE_LOG : NTT_TILE_LOG - 2;
    const uint32_t C_LOG = 6;  // strided passes move runs of 64 rows (256 B for Fp)
    // Stages left for strided passes, spread evenly over the fewest passes that fit a tile.
    const uint32_t r_max = tile_log - C_LOG;
    uint32_t strided = L > tile_log ? L - tile_log : 0;
    uint32_t n_strided = (strided + r_max - 1) / r_max;
    OVM_PROF(sizeof(T) == 4 ? "ntt.dif_passes" : "ntt.dif_passes_ext", stream);
    uint32_t s = 0;
    const T *src = in;
    uint64_t src_stride = in_col_stride;
    uint32_t src_log_n = log_n_in;
    while (s < L) {
        uint32_t remaining = L - s;
        uint32_t r, c;
        if (remaining <= tile_log) {
            r = remaining;  // final, contiguous pass
            c = 0;
        } else {
            c = C_LOG;
            r = (strided + n_strided - 1) / n_strided;
            strided -= r;
            n_strided--;
        }
        uint64_t tiles = 1ull << (L - r - c);
        dim3 grid((unsigned)tiles, (unsigned)width);
        size_t shmem = sizeof(T) << (r + c);
        OVM_LAUNCH(dif_pass_kernel<T>, grid, dim3(NTT_THREADS), shmem, stream, src, out, L,
                           src_log_n, s, r, c, src_stride, out_col_stride, *tw);
        OVM_CHECK_LAUNCH();
        s += r;
        src = out;
        src_stride = out_col_stride;
        src_log_n = L;
    }
    return 0;
}

}  // namespace ovm

using namespace ovm;

extern "C" {

int ovm_hip_eval_to_coeff(const uint32_t *d_in, uint32_t *d_out, uint64_t width,
                          uint32_t log_height, uint32_t l_skip, ovm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (l_skip > log_height || l_skip > NTT_TILE_LOG) return (int)hipErrorInvalidValue;
    if (width == 0) return 0;
    const Twiddles *tw = twiddles();
    if (!tw) return (int)hipErrorInvalidValue;
    OVM_PROF("ntt.eval_to_coeff", stream);
    if (l_skip == 4) {
        // columns are contiguous (column stride = 2^log_height), so the chunks of all columns form one array
        const uint64_t n_chunks = width << (log_height - 4);
        if (tuning().lde_transposed && (n_chunks & 255) == 0)
            OVM_LAUNCH(eval_to_coeff16_kernel<true>, dim3((unsigned)(n_chunks / 256)), dim3(256), 0, stream, (const Fp *)d_in,
                       (Fp *)d_out, n_chunks, Fp::from_canonical(16).inv());
        else
            OVM_LAUNCH(eval_to_coeff16_kernel<false>, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, stream,
                       (const Fp *)d_in, (Fp *)d_out, n_chunks, Fp::from_canonical(16).inv());
        OVM_CHECK_LAUNCH();
        return 0;
    }
    uint32_t tile_log = log_height < (uint32_t)NTT_TILE_LOG ? log_height : (uint32_t)NTT_TILE_LOG;
    dim3 grid(1u << (log_height - tile_log), (unsigned)width);
    Fp inv_chunk = Fp::from_canonical(1u << l_skip).inv();
    OVM_LAUNCH(eval_to_coeff_kernel, grid, dim3(NTT_THREADS), sizeof(Fp) << tile_log, stream,
                       (const Fp *)d_in, (Fp *)d_out, log_height, l_skip, tile_log, inv_chunk, *tw);
    OVM_CHECK_LAUNCH();
    return 0;
}

int ovm_hip_ntt_dif(const uint32_t *d_in, uint32_t *d_out, uint64_t width, uint32_t log_n,
                    uint32_t log_n_in, ovm_stream_t stream) {
    if (log_n_in > log_n) return (int)hipErrorInvalidValue;
    return dif_ntt<Fp>((const Fp *)d_in, (Fp *)d_out, width, log_n, log_n_in, 1ull << log_n_in,
                       1ull << log_n, (hipStream_t)stream);
}

int ovm_hip_ntt_dif_ext(const uint32_t *d_in, uint32_t *d_out, uint64_t width, uint32_t log_n,
                        uint32_t log_n_in, ovm_stream_t stream) {
    if (log_n_in > log_n) return (int)hipErrorInvalidValue;
    return dif_ntt<FpExt>((const FpExt *)d_in, (FpExt *)d_out, width, log_n, log_n_in,
                          1ull << log_n_in, 1ull << log_n, (hipStream_t)stream);
}

int ovm_hip_rs_encode(const uint32_t *d_stacked, uint32_t *d_scratch, uint32_t *d_codeword,
                      uint64_t width, uint32_t log_height, uint32_t l_skip, uint32_t log_blowup,
                      ovm_stream_t stream) {
    if (log_height + log_blowup > TW_LOG_MAX) return (int)hipErrorInvalidValue;
    int rc = ovm_hip_eval_to_coeff(d_stacked, d_scratch, width, log_height, l_skip, stream);
    if (rc) return rc;
    return dif_ntt<Fp>((const Fp *)d_scratch, (Fp *)d_codeword, width, log_height + log_blowup,
                       log_height, 1ull << log_height, 1ull << (log_height + log_blowup),
                       (hipStream_t)stream);
}

}  // extern "C"
