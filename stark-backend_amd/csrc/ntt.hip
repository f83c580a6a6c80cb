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
#include "device_ctx.h"
#include "openvm_hip.h"

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
        hipLaunchKernelGGL(dif_pass_kernel<T>, grid, dim3(NTT_THREADS), shmem, stream, src, out, L,
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
    uint32_t tile_log = log_height < (uint32_t)NTT_TILE_LOG ? log_height : (uint32_t)NTT_TILE_LOG;
    dim3 grid(1u << (log_height - tile_log), (unsigned)width);
    Fp inv_chunk = Fp::from_canonical(1u << l_skip).inv();
    hipLaunchKernelGGL(eval_to_coeff_kernel, grid, dim3(NTT_THREADS), sizeof(Fp) << tile_log, stream,
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
