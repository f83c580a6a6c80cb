// Kernels of the sumcheck stages: LogUp-GKR, zerocheck / logup batch sumcheck (round 0 on cosets
// and the multilinear rounds), stacked opening reduction and WHIR folding.
//
// Replaces (reference, all under crates/cuda-backend/cuda/src/): logup_zerocheck/gkr_input.cu,
// gkr.cu, zerocheck_round0.cu, logup_round0.cu, mle.cu, utils.cu (fold_ple), sumcheck.cu (folds,
// block reductions), stacked_reduction.cu, whir.cu, poly.cu (eq tables, Horner). Algorithms
// follow the protocol as specified by crates/stark-backend/src/prover/{sumcheck,stacked_reduction,
// whir}.rs and prover/logup_zerocheck/; the decomposition into kernels is this backend's own.
//
// Shape of every reduction kernel: a fixed grid of RED_BLOCKS x RED_THREADS lanes strides over
// the domain with unit-stride (coalesced) accesses, keeps its partial sums in registers, reduces
// them through wave shuffles and LDS, and writes one record per workgroup; `sum_partials`
// finishes. Field addition is exact, so the result does not depend on the schedule.
#include "kernels.h"
#include "eval.cuh"
#include "cached_vm.cuh"
#include "mle_vm.cuh"
#include "wave_chunks.cuh"

namespace ovm {

__device__ __forceinline__ EF wave_sum(EF v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        EF o;
#pragma unroll
        for (int c = 0; c < 4; c++) o.c[c] = Fp::raw(__shfl_down(v.c[c].v, off));
        v = v + o;
    }
    return v;
}

// Sum K values per lane over the workgroup; lanes 0..K-1 store them at out[0..K). With SYS the stores are released
// to the host (out is a host mailbox), otherwise -- if `fence` -- to the device (out is a partials record another
// workgroup will read): write-through stores and a wait for them, not a fence (device_common.h: release_to_*).
template <int K, bool SYS = false>
__device__ void block_sum_store(EF (&v)[K], EF *out, bool fence = true) {
    __shared__ EF sh[(RED_THREADS / 64) * K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        EF s = wave_sum(v[k]);
        if (lane == 0) sh[wave * K + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        EF s = sh[threadIdx.x];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) s = s + sh[w * K + threadIdx.x];
        if (SYS) {
            st_ef_sys(out + threadIdx.x, s);
            release_to_host();
        } else if (fence) {
            st_ef_agent(out + threadIdx.x, s);
            release_to_device();
        } else {
            st_ef(out + threadIdx.x, s);
        }
    }
    __syncthreads();
}

// Make everything this thread stored visible to the host, then raise the mailbox flag.
__device__ __forceinline__ void publish(const RedOut &ro) {
    if (ro.flag) {
        release_to_host();
        st_word_sys(ro.flag, ro.seq);
    }
}

// Tail of every reduction kernel ("last workgroup finishes"): each workgroup has stored its
// partial records ro.partials[b * K + k] through block_sum_store; the workgroup that arrives last
// at the counter sums the n_records records, writes the K totals to ro.out (a host-mapped mailbox)
// and raises the flag the host is spinning on. No second launch, no device-to-host copy, no
// stream synchronisation per sumcheck round. Field addition is exact and commutative, so the
// arrival order does not change the result.
template <int KMAX>
__device__ void finish_reduction(const RedOut &ro, uint32_t n_records, uint32_t K) {
    if (ro.counter == nullptr) return;  // large grid: sum_partials finishes (see red_finish)
    if (gridDim.x * gridDim.y * gridDim.z == 1) {  // block_sum_out wrote the totals themselves
        if (threadIdx.x == 0) publish(ro);
        return;
    }
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) {
        const uint32_t total = gridDim.x * gridDim.y * gridDim.z;
        s_last = total == 1 || atomicAdd(ro.counter, 1u) == total - 1;
    }
    __syncthreads();
    if (!s_last) return;
    acquire_from_device();
    EF acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) acc[k] = EF::zero();
    for (uint32_t b = threadIdx.x; b < n_records; b += RED_THREADS) {
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if ((uint32_t)k < K) acc[k] = acc[k] + ld_ef_agent(ro.partials + (size_t)b * K + k);
    }
    block_sum_store<KMAX, true>(acc, ro.out);  // KMAX totals (entries >= K are zero)
    if (threadIdx.x == 0) {
        *ro.counter = 0;
        publish(ro);
    }
}

// Large-K variant as its own launch: block k sums entry k of every record; the last block raises
// the flag.
__global__ void sum_partials_kernel(RedOut ro, uint32_t n_blocks, uint32_t K) {
    const uint32_t k = blockIdx.x;
    EF acc[1] = {EF::zero()};
    for (uint32_t b = threadIdx.x; b < n_blocks; b += blockDim.x) acc[0] = acc[0] + ld_ef(ro.partials + (size_t)b * K + k);
    block_sum_store<1, true>(acc, ro.out + k);
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) s_last = atomicAdd(ro.counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        *ro.counter = 0;
        publish(ro);
    }
}

int sum_partials(const RedOut &ro, uint32_t n_blocks, uint32_t K, hipStream_t s) {
    OVM_LAUNCH(sum_partials_kernel, dim3(K), dim3(RED_THREADS), 0, s, ro, n_blocks, K);
    OVM_CHECK_LAUNCH();
    return 0;
}

// Every agent-scope release on gfx950 writes the XCD's L2 back; with thousands of workgroups doing
// it the tail costs ~50 ns per workgroup (measured: +100 us on a 2048-workgroup launch). So the
// in-kernel finish is used for grids of at most tuning().finish_inline_max (64) workgroups -- the
// latency-bound rounds, which are most of them -- and larger grids take one extra sum_partials launch.
// (Round 2 tried storing the records through to memory instead -- agent-scope atomic stores + s_waitcnt, no
// release fence -- with the in-kernel finish for every grid: correct, 159 tests green, and no faster:
// 78.1 vs 77.3-78.1 ms per proof; the last workgroup's pass over 2048 records costs what the launch did.)
// (Round 3 built a third way, 64-bit atomic accumulators instead of records: faster in isolation, not in the prover --
// profiles/r03/finish_atomic_sweep.json, ubench_finish.json -- and removed in round 5.)
static inline RedOut red_inline(const RedOut &ro, uint32_t total_blocks, uint32_t K) {
    RedOut r = ro;
    if (total_blocks > tuning().finish_inline_max) r.counter = nullptr;
    return r;
}
static inline int red_finish(const RedOut &ro, uint32_t total_blocks, uint32_t n_records, uint32_t K, hipStream_t s) {
    if (total_blocks > tuning().finish_inline_max) return sum_partials(ro, n_records, K, s);
    return 0;
}
__device__ __forceinline__ bool solo_workgroup(const RedOut &ro) {
    return gridDim.x * gridDim.y * gridDim.z == 1 && ro.counter != nullptr;
}
// a workgroup's K totals as its record
template <int K>
__device__ __forceinline__ void block_sum_out(EF (&v)[K], const RedOut &ro, size_t record_off) {
    // (a launch of ONE workgroup -- most sub-rounds of the small GKR layers, the tails of every sumcheck -- has no
    // records to combine: its totals go straight to the mailbox, and finish_reduction only raises the flag)
    if (solo_workgroup(ro)) block_sum_store<K, true>(v, ro.out + record_off);
    else block_sum_store<K>(v, ro.partials + record_off, ro.counter != nullptr);
}

// The job a workgroup of a batched launch belongs to: the last one whose first index is <= v (the jobs' index ranges are
// contiguous and ascending). A binary search: a 46-AIR proof batches ~90 jobs, and the linear scan this replaces was
// up to 90 dependent scalar loads before a workgroup of the last jobs did anything.
template <class First>
__device__ __forceinline__ uint32_t find_job(uint32_t n_jobs, uint32_t v, First first) {
    uint32_t lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (first(mid) <= v) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

static inline uint32_t red_grid(uint64_t work_items) {
    uint64_t b = (work_items + RED_THREADS - 1) / RED_THREADS;
    if (b < 1) b = 1;
    return (uint32_t)(b > RED_BLOCKS ? RED_BLOCKS : b);
}

// ---- selectors and eq tables -----------------------------------------------------------------
__global__ void build_selectors_kernel(Fp *sels, uint32_t log_lifted, uint32_t log_height) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lifted = 1ull << log_lifted, h = 1ull << log_height;
    if (p >= lifted) return;
    uint64_t i = p & (h - 1);
    sels[p] = i == 0 ? Fp::one() : Fp::zero();
    sels[lifted + p] = i == h - 1 ? Fp::zero() : Fp::one();
    sels[2 * lifted + p] = i == h - 1 ? Fp::one() : Fp::zero();
}
int build_selectors(Fp *sels, uint32_t log_lifted, uint32_t log_height, hipStream_t s) {
    uint64_t n = 1ull << log_lifted;
    OVM_LAUNCH(build_selectors_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, sels, log_lifted, log_height);
    OVM_CHECK_LAUNCH();
    return 0;
}

// Output policy: 16-byte AoS elements (the GKR phase builds its own limb-planar eq heap).
struct EqOutAos {
    EF *p;
    __device__ EF ld(uint64_t i) const { return ld_ef(p + i); }
    __device__ void st(uint64_t i, const EF &x) const { st_ef(p + i, x); }
};
template <bool MOBIUS, class Out>
__global__ void eq_table_kernel(const EF *point, uint32_t n, Out out) {
    uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= (1ull << n)) return;
    EF acc = EF::one();
    for (uint32_t i = 0; i < n; i++) {
        EF x = ld_ef(point + i);
        EF f;
        if (MOBIUS)
            f = (y >> i & 1) ? x : EF::one() - x.dbl();
        else
            f = (y >> i & 1) ? x : EF::one() - x;
        acc = acc * f;
    }
    out.st(y, acc);
}
// One doubling step: entries [0, 2^k) hold the table over variables 0..k-1.
//   eq:     out[y + 2^k] = out[y] * x_k ;  out[y] = out[y] - out[y + 2^k]          (= out[y] * (1 - x_k))
//   mobius: out[y + 2^k] = out[y] * x_k ;  out[y] = out[y] - 2 * out[y + 2^k]      (= out[y] * (1 - 2 x_k))
// One extension multiply per pair, so a 2^n table costs 2^n multiplies in total instead of n * 2^n.
template <bool MOBIUS, class Out>
__global__ void eq_double_kernel(const EF *point, uint32_t k, Out out) {
    uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= (1ull << k)) return;
    EF lo = out.ld(y);
    EF hi = lo * ld_ef(point + k);
    out.st(y + (1ull << k), hi);
    out.st(y, MOBIUS ? lo - hi.dbl() : lo - hi);
}

static constexpr uint32_t EQ_DIRECT_BITS = 16;  // (a launch is ~4.5 us, 2^16 entries x 16 products are ~2 us of the device: the MLE rounds built their tables with up to 8 doubling launches each)

template <bool MOBIUS, class Out>
static int eq_table_impl(const EF *d_point, uint32_t n, Out out, hipStream_t s) {
    uint32_t n0 = n < EQ_DIRECT_BITS ? n : EQ_DIRECT_BITS;
    OVM_LAUNCH((eq_table_kernel<MOBIUS, Out>), dim3(ceil_div(1ull << n0, 256)), dim3(256), 0, s, d_point, n0,
                       out);
    OVM_CHECK_LAUNCH();
    for (uint32_t k = n0; k < n; k++) {
        OVM_LAUNCH((eq_double_kernel<MOBIUS, Out>), dim3(ceil_div(1ull << k, 256)), dim3(256), 0, s, d_point,
                           k, out);
        OVM_CHECK_LAUNCH();
    }
    return 0;
}
int eq_table(const EF *d_point, uint32_t n, EF *out, hipStream_t s) {
    return eq_table_impl<false>(d_point, n, EqOutAos{out}, s);
}
int mobius_eq_table(const EF *d_point, uint32_t n, EF *out, hipStream_t s) {
    return eq_table_impl<true>(d_point, n, EqOutAos{out}, s);
}

// ---- coset LDE of 2^l_skip-row chunks --------------------------------------------------------
static constexpr int LDE_THREADS = 256;
static constexpr uint32_t LDE_TILE_LOG = 11;

__global__ __launch_bounds__(LDE_THREADS) void coset_lde_kernel(const ColDesc *cols, const uint32_t *used,
                                                                uint32_t n_cols_total, uint32_t log_lifted,
                                                                uint32_t l_skip, uint32_t tile_log,
                                                                uint32_t n_cosets, const Fp *shift_pows,
                                                                Fp inv_n, Fp *lde, Twiddles tw, uint64_t row0) {
    extern __shared__ Fp lds[];
    const uint32_t tile = 1u << tile_log;
    Fp *coef = lds, *work = lds + tile;
    const uint32_t col = used[blockIdx.y];
    const ColDesc cd = cols[col];
    const uint64_t base = (uint64_t)blockIdx.x << tile_log;
    const uint64_t hmask = (1ull << cd.log_height) - 1;
    for (uint32_t i = threadIdx.x; i < tile; i += LDE_THREADS) coef[i] = cd.ptr[(row0 + base + i + cd.rot) & hmask];
    __syncthreads();
    const uint32_t half = tile >> 1, N = 1u << l_skip;
    // inverse DIF over the chunk bits: natural -> bit-reversed coefficients
    for (uint32_t k = 0; k < l_skip; k++) {
        const uint32_t bit = l_skip - 1 - k, lo_mask = (1u << bit) - 1;
        for (uint32_t bf = threadIdx.x; bf < half; bf += LDE_THREADS) {
            uint32_t i0 = ((bf & ~lo_mask) << 1) | (bf & lo_mask), i1 = i0 | (1u << bit);
            uint32_t e = i0 & lo_mask;
            uint32_t e27 = (uint32_t)(((1u << TW_LOG_MAX) - ((uint64_t)e << (TW_LOG_MAX - bit - 1))) & ((1u << TW_LOG_MAX) - 1));
            Fp a = coef[i0], b = coef[i1];
            coef[i0] = a + b;
            Fp d = a - b;
            coef[i1] = e ? d * tw_pow27(tw, e27) : d;
        }
        __syncthreads();
    }
    for (uint32_t c = 0; c < n_cosets; c++) {
        // twist: coefficient of degree i (stored at bitrev position) times shift^i / N
        for (uint32_t i = threadIdx.x; i < tile; i += LDE_THREADS) {
            uint32_t deg = bitrev32(i & (N - 1), l_skip);
            work[i] = coef[i] * (shift_pows[c * N + deg] * inv_n);
        }
        __syncthreads();
        // forward DIT: bit-reversed coefficients -> natural evaluations
        for (uint32_t bit = 0; bit < l_skip; bit++) {
            const uint32_t lo_mask = (1u << bit) - 1;
            for (uint32_t bf = threadIdx.x; bf < half; bf += LDE_THREADS) {
                uint32_t i0 = ((bf & ~lo_mask) << 1) | (bf & lo_mask), i1 = i0 | (1u << bit);
                uint32_t e = i0 & lo_mask;
                Fp a = work[i0], b = work[i1];
                if (e) b = b * tw_pow27(tw, e << (TW_LOG_MAX - bit - 1));
                work[i0] = a + b;
                work[i1] = a - b;
            }
            __syncthreads();
        }
        Fp *dst = lde + (((uint64_t)c * n_cols_total + col) << log_lifted) + base;
        for (uint32_t i = threadIdx.x; i < tile; i += LDE_THREADS) dst[i] = work[i];
        __syncthreads();
    }
}

// Register form for 16-row chunks (l_skip = 4, the production parameter): a lane owns one chunk.
// Every twiddle is a compile-time constant (powers of omega_16), so the chunk costs 17 multiplies
// for the inverse transform and 16 + 17 per coset instead of 32 generic ones per transform with a
// table lookup each, and there is no LDS traffic or barrier at all.
#define W16P(j) bb::cx_to_mont(bb::cx_powmod(bb::TWO_ADIC_GENERATORS[4], (j)))
static constexpr uint32_t LDE_W16[8] = {W16P(0), W16P(1), W16P(2), W16P(3), W16P(4), W16P(5), W16P(6), W16P(7)};
static constexpr uint32_t LDE_W16_INV[8] = {W16P(0),  W16P(15), W16P(14), W16P(13),
                                            W16P(12), W16P(11), W16P(10), W16P(9)};
#undef W16P
__device__ __forceinline__ constexpr int bitrev4(int i) {
    return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
}

// A list entry with bit 31 set also produces the column's ROTATED view (column twins[col]) from the same registers:
// the rotated chunk is the chunk's polynomial at omega * X, corrected in its last point -- it holds T(x + 1, 0) where
// P(omega X) holds T(x, 0) again -- so on the coset point s * omega^z
//     rot(z) = plain((z + 1) mod 16) + (T(x + 1, 0) - T(x, 0)) * L_15(s * omega^z),
// L_15 the Lagrange polynomial of the chunk's last point (lag[c * 16 + z], host-computed): 16 multiply-adds per coset
// instead of a second interpolation and extension from 16 unaligned loads.
// XPOSE (tall traces: every wave owns 64 consecutive chunks = 4 KiB of the column): the chunks change lanes in LDS so
// that the global loads and stores of a wave are whole 1 KiB runs (wave_chunks.cuh).
template <bool XPOSE>
__global__ __launch_bounds__(256) void coset_lde16_kernel(const ColDesc *__restrict__ cols, const uint32_t *__restrict__ used,
                                                          const int32_t *__restrict__ twins, uint32_t n_cols_total,
                                                          uint32_t log_lifted, uint32_t n_cosets, const Fp *__restrict__ twist,
                                                          const Fp *__restrict__ lag, Fp *__restrict__ lde, uint64_t row0) {
    __shared__ uint4 tile[XPOSE ? 4 * WAVE_CHUNK_TILE : 1];
    const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!XPOSE && chunk >= (1ull << (log_lifted - 4))) return;
    const uint32_t entry = used[blockIdx.y], col = entry & 0x7fffffffu;
    const bool with_twin = entry >> 31;
    const ColDesc cd = cols[col];
    // (row0: the slab of rows [row0, row0 + 2^log_lifted) of the columns is extended -- round 0 in slabs)
    const uint64_t hmask = (1ull << cd.log_height) - 1, base = chunk << 4, sbase = base + row0;
    const uint32_t lane = threadIdx.x & 63;
    uint4 *T = tile + (XPOSE ? (threadIdx.x >> 6) * WAVE_CHUNK_TILE : 0);
    const uint64_t wave_base = (chunk - lane) << 4;  // first row of the wave's 64 chunks (XPOSE)
    auto xpose_out = [&](uint4 *dst, const Fp (&y)[16]) {
        uint4 own[4];
#pragma unroll
        for (int v = 0; v < 4; v++) own[v] = make_uint4(y[4 * v].v, y[4 * v + 1].v, y[4 * v + 2].v, y[4 * v + 3].v);
        wave_chunks_store(T, dst, lane, own);
    };
    Fp x[16];
    if (XPOSE && cd.rot == 0) {
        uint4 own[4];
        wave_chunks_load(T, reinterpret_cast<const uint4 *>(cd.ptr + ((wave_base + row0) & hmask)), lane, own);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            x[4 * v] = Fp::raw(own[v].x);
            x[4 * v + 1] = Fp::raw(own[v].y);
            x[4 * v + 2] = Fp::raw(own[v].z);
            x[4 * v + 3] = Fp::raw(own[v].w);
        }
    } else if (cd.rot == 0 && cd.log_height >= 4) {
        const uint4 *src = reinterpret_cast<const uint4 *>(cd.ptr + (sbase & hmask));
#pragma unroll
        for (int v = 0; v < 4; v++) {
            uint4 q = src[v];
            x[4 * v] = Fp::raw(q.x);
            x[4 * v + 1] = Fp::raw(q.y);
            x[4 * v + 2] = Fp::raw(q.z);
            x[4 * v + 3] = Fp::raw(q.w);
        }
    } else {
#pragma unroll
        for (int z = 0; z < 16; z++) x[z] = cd.ptr[(sbase + z + cd.rot) & hmask];
    }
    Fp delta = Fp::zero();
    uint32_t twin_col = 0;
    if (with_twin) {
        delta = cd.ptr[(sbase + 16) & hmask] - x[0];
        twin_col = (uint32_t)twins[col];
    }
    // inverse DIF, natural -> bit-reversed coefficients (unscaled; 1/16 is folded into `twist`)
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
                x[g + j + half] = j == 0 ? d : d * Fp::raw(LDE_W16_INV[j << (3 - bit)]);
            }
        }
    }
    for (uint32_t c = 0; c < n_cosets; c++) {
        Fp y[16];
        // twist: the coefficient of degree bitrev(i) sits in register i
#pragma unroll
        for (int i = 0; i < 16; i++) y[i] = x[i] * twist[c * 16 + bitrev4(i)];
        // forward DIT, bit-reversed coefficients -> natural evaluations
#pragma unroll
        for (int bit = 0; bit < 4; bit++) {
            const int half = 1 << bit;
#pragma unroll
            for (int g = 0; g < 16; g += 2 * half) {
#pragma unroll
                for (int j = 0; j < half; j++) {
                    Fp a = y[g + j], b = y[g + j + half];
                    if (j) b = b * Fp::raw(LDE_W16[j << (3 - bit)]);
                    y[g + j] = a + b;
                    y[g + j + half] = a - b;
                }
            }
        }
        Fp *const out_col = lde + (((uint64_t)c * n_cols_total + col) << log_lifted);
        if (XPOSE) {
            xpose_out(reinterpret_cast<uint4 *>(out_col + wave_base), y);
        } else {
            uint4 *dst = reinterpret_cast<uint4 *>(out_col + base);
#pragma unroll
            for (int v = 0; v < 4; v++) dst[v] = make_uint4(y[4 * v].v, y[4 * v + 1].v, y[4 * v + 2].v, y[4 * v + 3].v);
        }
        if (with_twin) {
            Fp w[16];
#pragma unroll
            for (int z = 0; z < 16; z++) w[z] = y[(z + 1) & 15] + delta * lag[c * 16 + z];
            Fp *const out_twin = lde + (((uint64_t)c * n_cols_total + twin_col) << log_lifted);
            if (XPOSE) {
                xpose_out(reinterpret_cast<uint4 *>(out_twin + wave_base), w);
            } else {
                uint4 *dst2 = reinterpret_cast<uint4 *>(out_twin + base);
#pragma unroll
                for (int v = 0; v < 4; v++) dst2[v] = make_uint4(w[4 * v].v, w[4 * v + 1].v, w[4 * v + 2].v, w[4 * v + 3].v);
            }
        }
    }
}

int coset_lde(const ColDesc *d_cols, const uint32_t *d_used, uint32_t n_used, uint32_t n_cols_total,
              uint32_t log_lifted, uint32_t l_skip, uint32_t n_cosets, const Fp *d_shift_pows, Fp *lde,
              hipStream_t s, uint64_t row0, uint32_t pad_lds_bytes, const int32_t *d_twins, const Fp *d_lag) {
    if (n_used == 0 || n_cosets == 0) return 0;
    if (row0 & ((1ull << l_skip) - 1)) return (int)hipErrorInvalidValue;
    const Twiddles *tw = twiddles();
    if (!tw || l_skip > LDE_TILE_LOG || l_skip > log_lifted) return (int)hipErrorInvalidValue;
    // d_shift_pows[k * N + i] = shift_k^i / N (the caller folds 1/N in)
    if (l_skip == 4) {
        dim3 grid16((unsigned)ceil_div(1ull << (log_lifted - 4), 256), n_used);
        // (through LDS when every wave owns 64 whole chunks inside one period of the column: 2^12 rows up, slabs of 2^10)
        // (log_lifted > l_skip: the columns are at least as tall as the rows extended here)
        const bool xpose = tuning().lde_transposed && log_lifted >= 12 && (row0 & 1023) == 0;
        const unsigned dyn = pad_lds_bytes > 65536 ? 65536 : pad_lds_bytes;
        if (xpose)
            OVM_LAUNCH(coset_lde16_kernel<true>, grid16, dim3(256), dyn, s, d_cols, d_used, d_twins, n_cols_total, log_lifted, n_cosets,
                       d_shift_pows, d_lag, lde, row0);
        else
            OVM_LAUNCH(coset_lde16_kernel<false>, grid16, dim3(256), dyn, s, d_cols, d_used, d_twins, n_cols_total, log_lifted, n_cosets,
                       d_shift_pows, d_lag, lde, row0);
        OVM_CHECK_LAUNCH();
        return 0;
    }
    uint32_t tile_log = log_lifted < LDE_TILE_LOG ? log_lifted : LDE_TILE_LOG;
    dim3 grid(1u << (log_lifted - tile_log), n_used);
    OVM_LAUNCH(coset_lde_kernel, grid, dim3(LDE_THREADS), 2 * sizeof(Fp) << tile_log, s, d_cols, d_used,
                       n_cols_total, log_lifted, l_skip, tile_log, n_cosets, d_shift_pows, Fp::one(), lde, *tw, row0);
    OVM_CHECK_LAUNCH();
    return 0;
}

// ---- round 0 evaluation ----------------------------------------------------------------------
struct LdeFetch {
    const Fp *base;  // lde of coset k, column 0
    uint32_t log_lifted;
    uint64_t p;
    __device__ Fp operator()(uint32_t col) const { return base[((uint64_t)col << log_lifted) + p]; }
};

// The interpreter evaluates ROWS rows per lane per pass over the program: decode, branches and
// scalar work (the larger part of the issue slots, see below) are paid once per ROWS field
// operations. round0_eval picks ROWS from the trace height so the grid still fills the chip.
// What bounds the interpreter (SQ counters on the Keccak-shaped 9.5K-instruction program, 2^20 rows):
// not memory -- pointing every column operand at 8 cached columns changes nothing -- but issue slots:
// ~37 scalar and ~30 vector instructions per interpreted instruction, plus 8-way LDS bank conflicts
// where the compiler had placed the per-lane slot array. Hence:
//  * one 16-byte scalar load per instruction and shift/mask decoding (no per-field loads);
//  * blocks of R0_PREFETCH instructions: all column operands of a block are requested, then it runs
//    (the loads are otherwise serially dependent: decode -> load -> use, one at a time);
//  * slots in LDS as [slot][row][lane] words (conflict-free) while they fit, else private memory;
//  * ACC adds raw 62-bit products into 96-bit accumulators (v_mad_u64_u32 + carry) and reduces
//    once per row instead of four Montgomery reductions and four modular adds per ACC.
// (Acc96 / mac96 / mac_weight: eval.cuh -- shared with the run-time compiled kernels)

template <int NS, int R, int NOUT>
struct InterpRound0 {
    static constexpr int ROWS = R;
    static constexpr bool REDUCE_IN_IMAGE = false;
    static constexpr int PREFETCH = 8 / ROWS;
    static constexpr bool LDS_SLOTS = NS * ROWS <= 32;  // <= 32 KiB per workgroup
    DevProgram prog;

    struct Slots {
        uint32_t *lds;
        uint32_t priv[LDS_SLOTS ? 1 : NS * ROWS];
        __device__ __forceinline__ Fp get(uint32_t s, int r) const {
            if constexpr (LDS_SLOTS) return Fp::raw(lds[(s * ROWS + r) * RED_THREADS]);
            else return Fp::raw(priv[s * ROWS + r]);
        }
        __device__ __forceinline__ void set(uint32_t s, int r, Fp v) {
            if constexpr (LDS_SLOTS) lds[(s * ROWS + r) * RED_THREADS] = v.v;
            else priv[s * ROWS + r] = v.v;
        }
    };
    struct Decoded {
        uint32_t op, ak, bk, acc, dst, a, b;
        __device__ __forceinline__ explicit Decoded(const uint4 raw)
            : op(raw.x & 0xff), ak((raw.x >> 8) & 0xff), bk((raw.x >> 16) & 0xff), acc(raw.x >> 24),
              dst(raw.y & 0xffff), a(raw.z), b(raw.w) {}
    };
    static __device__ __forceinline__ Fp operand(uint32_t kind, uint32_t payload, const Slots &slots, int r, Fp col) {
        return kind == K_COL ? col : kind == K_SLOT ? slots.get(payload, r) : Fp::raw(payload);
    }
    static __device__ __forceinline__ void exec(const Decoded &in, const EF *weights, Slots &slots, const Fp (&ca)[ROWS],
                                                const Fp (&cb)[ROWS], Acc96 (&acc)[ROWS][NOUT]) {
        Fp a[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) a[r] = operand(in.ak, in.a, slots, r, ca[r]);
        switch (in.op) {
            case OP_ADD:
#pragma unroll
                for (int r = 0; r < ROWS; r++) slots.set(in.dst, r, a[r] + operand(in.bk, in.b, slots, r, cb[r]));
                break;
            case OP_SUB:
#pragma unroll
                for (int r = 0; r < ROWS; r++) slots.set(in.dst, r, a[r] - operand(in.bk, in.b, slots, r, cb[r]));
                break;
            case OP_MUL:
#pragma unroll
                for (int r = 0; r < ROWS; r++) slots.set(in.dst, r, a[r] * operand(in.bk, in.b, slots, r, cb[r]));
                break;
            case OP_NEG:
#pragma unroll
                for (int r = 0; r < ROWS; r++) slots.set(in.dst, r, -a[r]);
                break;
            default: {  // OP_ACC
                const EF w = weights[in.b];
                if (NOUT > 1 && in.acc) {
#pragma unroll
                    for (int r = 0; r < ROWS; r++) acc[r][NOUT - 1].add(w, a[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < ROWS; r++) acc[r][0].add(w, a[r]);
                }
                break;
            }
        }
    }
    __device__ __forceinline__ void operator()(const Fp *lde_k, uint32_t log_lifted, const uint64_t (&p)[ROWS],
                                               EF (&out)[ROWS][2]) const {
        __shared__ uint32_t lds_slots[LDS_SLOTS ? NS * ROWS * RED_THREADS : 1];
        Slots slots;
        slots.lds = lds_slots + threadIdx.x;
        Acc96 acc[ROWS][NOUT];
#pragma unroll
        for (int r = 0; r < ROWS; r++)
#pragma unroll
            for (int o = 0; o < NOUT; o++) acc[r][o].clear();
        const uint4 *code = reinterpret_cast<const uint4 *>(prog.code);
        const uint32_t n_blocked = prog.n_instr - prog.n_instr % PREFETCH;
        for (uint32_t pc = 0; pc < n_blocked; pc += PREFETCH) {
            uint4 raw[PREFETCH];
            Fp ca[PREFETCH][ROWS], cb[PREFETCH][ROWS];
#pragma unroll
            for (int d = 0; d < PREFETCH; d++) {
                raw[d] = code[pc + d];
                const Decoded in(raw[d]);
                if (in.ak == K_COL) {
#pragma unroll
                    for (int r = 0; r < ROWS; r++) ca[d][r] = lde_k[((uint64_t)in.a << log_lifted) + p[r]];
                }
                if (in.bk == K_COL && in.op <= OP_MUL) {
#pragma unroll
                    for (int r = 0; r < ROWS; r++) cb[d][r] = lde_k[((uint64_t)in.b << log_lifted) + p[r]];
                }
            }
#pragma unroll
            for (int d = 0; d < PREFETCH; d++) exec(Decoded(raw[d]), prog.weights, slots, ca[d], cb[d], acc);
        }
        for (uint32_t pc = n_blocked; pc < prog.n_instr; pc++) {
            const Decoded in(code[pc]);
            Fp ca[ROWS], cb[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                ca[r] = in.ak == K_COL ? lde_k[((uint64_t)in.a << log_lifted) + p[r]] : Fp::zero();
                cb[r] = in.bk == K_COL && in.op <= OP_MUL ? lde_k[((uint64_t)in.b << log_lifted) + p[r]] : Fp::zero();
            }
            exec(in, prog.weights, slots, ca, cb, acc);
        }
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
#pragma unroll
            for (int o = 0; o < NOUT; o++) out[r][o] = out[r][o] + acc[r][o].finish();
        }
    }
};
template <int NS, int R, int NOUT>
__global__ __launch_bounds__(RED_THREADS) void round0_eval_kernel(DevProgram prog, const Fp *lde,
                                                                  uint32_t n_cols_total, uint32_t log_lifted,
                                                                  uint32_t l_skip, uint32_t n_cosets,
                                                                  uint32_t n_out, const EF *eq_xi, EF add_const,
                                                                  EF *partials) {
    round0_eval_body(InterpRound0<NS, R, NOUT>{prog}, lde, n_cols_total, log_lifted, l_skip, n_cosets, n_out, eq_xi,
                     add_const, partials);
}

// ---- cached-column round-0 interpreter (program.h: stage_cached; cached_vm.cuh) -----------------
template <int R, int NOUT>
__global__ __launch_bounds__(RED_THREADS) void round0_cached_kernel(DevCached prog, const Fp *lde,
                                                                    uint32_t n_cols_total, uint32_t log_lifted,
                                                                    uint32_t l_skip, uint32_t n_cosets, uint32_t n_out,
                                                                    const EF *eq_xi, EF add_const, EF *partials) {
    round0_eval_body(CachedRound0<R, NOUT>{prog}, lde, n_cols_total, log_lifted, l_skip, n_cosets, n_out,
                     eq_xi, add_const, partials);
}

#define DISPATCH_SLOTS(NSLOTS, CALL)                      \
    do {                                                  \
        if ((NSLOTS) <= 8) { CALL(8); }                   \
        else if ((NSLOTS) <= 32) { CALL(32); }            \
        else if ((NSLOTS) <= 128) { CALL(128); }          \
        else if ((NSLOTS) <= 512) { CALL(512); }          \
        else return (int)hipErrorNotSupported;            \
    } while (0)

// Rows per lane: as many as leave >= 2^18 lanes (4 waves per SIMD), at most 4 accumulator sets per lane.
// OVM_R0_ROWS=1|2|4 overrides the choice (parity tests run every variant on small traces).
static int round0_rows(uint32_t log_lifted, uint32_t n_out) {
    int rows = log_lifted >= 20 ? 4 : log_lifted == 19 ? 2 : 1;
    if (const int v = tuning().r0_rows; v == 1 || v == 2 || v == 4) rows = v;
    while (rows > 1 && (rows * n_out > 4 || (1ull << log_lifted) / rows < RED_THREADS)) rows >>= 1;
    return rows;
}

// Layout policy of the cached-column interpreter (round 0): rows per lane R as above; the image is ONE pool of rows for
// cached columns and live values, R KiB per row, and it is sized for FOUR workgroups per CU (40 KiB): on the real Keccak-f
// program (2^20 rows, R = 4: 1 024 workgroups, exactly four per CU) 8 / 9 / 10 rows take 21.9 / 21.9 / 21.5 ms and 12 or 16
// rows -- three or two resident workgroups, a second wave of workgroups over the chip -- 34 ms; R = 2 takes 30-32 ms at
// every image from 16 to 32 KiB and R = 1 55 ms: with two rows per lane the scalar work per cell, not the wait for LDS,
// is what is left (profiles/r05/round0_cached_layout_sweep.txt). More rows than the program needs buy fewer column
// loads (7 100 per row with 4 spare rows, 5 400 with 6, 5 000 with 8).
static constexpr uint32_t R0_IMAGE_KIB = 40, R0_IMAGE_ROWS_MAX = 12;
CachedPlan round0_cached_plan(uint32_t log_lifted, uint32_t n_out, uint32_t min_rows, uint32_t /*n_instr*/) {
    CachedPlan plan;
    const Tuning &tn = tuning();  // (tuning / parity-test overrides)
    const uint32_t budget_kib = tn.stage_kib ? tn.stage_kib : R0_IMAGE_KIB;
    for (int rows = round0_rows(log_lifted, n_out); rows >= 1; rows >>= 1) {
        // (one row per lane may take what is left of a CU's LDS: better one resident workgroup than the plain interpreter)
        const uint32_t budget_rows = (rows > 1 ? budget_kib : std::max(budget_kib, 128u)) / (uint32_t)rows;
        if (budget_rows < min_rows) continue;
        plan.rows = (uint32_t)rows;
        plan.image_rows = tn.stage_rows ? std::min(budget_rows, std::max<uint32_t>(min_rows, (uint32_t)tn.stage_rows))
                                        : std::max(min_rows, std::min(budget_rows, R0_IMAGE_ROWS_MAX));
        plan.max_loads = tn.acc_rows ? (uint32_t)std::min<int>(std::max(2, tn.acc_rows), (int)CachedProgram::MAX_LOADS) : CachedProgram::MAX_LOADS;
        return plan;
    }
    return plan;  // rows == 0: does not fit, use the plain interpreter
}

// The leaf program: one row per lane; a small image (the programs are a handful of instructions per interaction) so that
// the launch keeps eight workgroups per CU -- it streams 4 B per cell in and 20 B per leaf out.
CachedPlan leaves_cached_plan(uint32_t min_rows) {
    CachedPlan plan;
    const Tuning &tn = tuning();
    const uint32_t budget_rows = tn.stage_kib ? tn.stage_kib : 64;
    if (budget_rows < min_rows) return plan;  // rows == 0: the plain interpreter
    plan.rows = 1;
    plan.image_rows = tn.stage_rows ? std::min(budget_rows, std::max<uint32_t>(min_rows, (uint32_t)tn.stage_rows))
                                    : std::max(min_rows, std::min(budget_rows, R0_IMAGE_ROWS_MAX));
    plan.max_loads = tn.acc_rows ? (uint32_t)std::min(std::max(2, tn.acc_rows), 4) : 4;
    return plan;
}

uint32_t round0_cached_grid(const DevCached &prog, uint32_t log_lifted) {
    const size_t lds_bytes = (size_t)prog.num_rows * prog.rows * RED_THREADS * 4;
    uint64_t lanes = (1ull << log_lifted) / prog.rows;
    if (lanes < RED_THREADS) lanes = RED_THREADS;
    // as many workgroups as are resident at once (LDS-bound), and every lane the same number of passes
    const uint32_t resident = 256u * (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (160u << 10) / std::max<size_t>(lds_bytes, 4096)));
    const uint32_t needed = (uint32_t)((lanes + RED_THREADS - 1) / RED_THREADS);
    const uint32_t passes = (needed + resident - 1) / resident;
    return (needed + passes - 1) / passes;
}

template <int R, int NOUT>
__global__ __launch_bounds__(RED_THREADS) void round0_cached_batched_kernel(const R0Job *jobs, uint32_t n_jobs,
                                                                            uint32_t l_skip, uint32_t *counters,
                                                                            RedOut ro) {
    uint32_t j = 0;
    j = find_job(n_jobs, blockIdx.x, [&](uint32_t i) { return jobs[i].wg0; });
    const R0Job job = jobs[j];
    const uint32_t bx = blockIdx.x - job.wg0;
    const uint32_t K = (job.n_cosets * NOUT) << l_skip;
    if (job.n_groups > 1) {
        const uint32_t grp = bx / job.gx_group;
#if defined(__HIP_DEVICE_COMPILE__)  // (scalar load of the group's descriptor: it feeds scalar operands of the VM)
        const DevCached gprog = ((const OVM_CONST_AS DevCached *)job.progs)[grp];
#else
        const DevCached gprog = job.progs[grp];
#endif
        round0_eval_body(CachedRound0<R, NOUT>{gprog}, job.lde, job.n_cols_total, job.log_lifted, l_skip,
                         job.n_cosets, NOUT, job.eq_xi, grp == 0 ? job.add_const : EF::zero(),
                         job.partials + (size_t)grp * job.gx_group * K, bx - grp * job.gx_group, job.gx_group, true);
    } else {
        round0_eval_body(CachedRound0<R, NOUT>{job.prog}, job.lde, job.n_cols_total, job.log_lifted, l_skip, job.n_cosets,
                         NOUT, job.eq_xi, job.add_const, job.partials, bx, job.gx, true);
    }
    // the job's last workgroup sums the job's records; the last job raises the flag
    __shared__ uint32_t s_last;
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(counters + 1 + j, 1u) == job.gx - 1;
    __syncthreads();
    if (!s_last) return;
    acquire_from_device();
    for (uint32_t k = threadIdx.x; k < K; k += RED_THREADS) {
        EF acc = EF::zero();
        for (uint32_t b = 0; b < job.gx; b++) acc = acc + ld_ef_agent(job.partials + (size_t)b * K + k);
        st_ef_sys(job.out + k, acc);
    }
    release_to_host();
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[1 + j] = 0;
        if (atomicAdd(counters, 1u) == n_jobs - 1) {
            counters[0] = 0;
            publish(ro);
        }
    }
}

int round0_eval_batched(const R0Job *d_jobs, const R0Job *h_jobs, uint32_t n_jobs, uint32_t rows, uint32_t n_out,
                        uint32_t l_skip, uint32_t *d_counters, const RedOut &ro, hipStream_t s) {
    if (n_jobs == 0) return 0;
    if (n_out < 1 || n_out > 2 || (1u << l_skip) > RED_THREADS) return (int)hipErrorInvalidValue;
    uint32_t grid = 0;
    size_t lds_bytes = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        const R0Job &job = h_jobs[j];
        if (!job.prog.words || job.prog.rows != rows || job.wg0 != grid || job.gx == 0 || job.n_cosets == 0 ||
            job.n_cosets > MAX_COSETS || job.n_groups < 1 || (job.n_groups > 1 && (!job.progs || job.gx != job.gx_group * job.n_groups)) ||
            job.lds_rows < job.prog.num_rows)
            return (int)hipErrorInvalidValue;
        grid += job.gx;
        lds_bytes = std::max(lds_bytes, (size_t)job.lds_rows * rows * RED_THREADS * 4);
    }
    lds_bytes = std::max(lds_bytes, sizeof(EF) * RED_THREADS);  // (the image doubles as the reduction's scratch)
#define LAUNCH(R, NOUT)                                                                                                 \
    OVM_LAUNCH((round0_cached_batched_kernel<R, NOUT>), dim3(grid), dim3(RED_THREADS), lds_bytes, s, d_jobs, n_jobs, \
                       l_skip, d_counters, ro)
    if (n_out == 1) {
        if (rows == 4) LAUNCH(4, 1);
        else if (rows == 2) LAUNCH(2, 1);
        else if (rows == 1) LAUNCH(1, 1);
        else return (int)hipErrorInvalidValue;
    } else {
        if (rows == 2) LAUNCH(2, 2);
        else if (rows == 1) LAUNCH(1, 2);
        else return (int)hipErrorInvalidValue;
    }
#undef LAUNCH
    OVM_CHECK_LAUNCH();
    return 0;
}

int round0_eval(const DevProgram &prog, const DevCached &staged, const Fp *lde, uint32_t n_cols_total,
                uint32_t log_lifted, uint32_t l_skip, uint32_t n_cosets, uint32_t n_out, const EF *d_eq_xi,
                EF add_const, const RedOut &ro, hipStream_t s) {
    if (n_cosets == 0) return 0;
    if (n_cosets > MAX_COSETS || n_out < 1 || n_out > 2 || (1u << l_skip) > 4 * RED_THREADS || log_lifted < l_skip)
        return (int)hipErrorInvalidValue;
    // a skip domain larger than a workgroup (l_skip 9, 10): the grid is a multiple of N / RED_THREADS (eval.cuh)
    const uint32_t wg_per_domain = (1u << l_skip) > RED_THREADS ? (1u << l_skip) / RED_THREADS : 1;
    auto whole_domains = [&](uint32_t g) { return (g + wg_per_domain - 1) / wg_per_domain * wg_per_domain; };
    int rows = round0_rows(log_lifted, n_out);
    if (staged.words) {
        rows = (int)staged.rows;  // fixed when the program was staged (round0_cached_plan)
        const size_t lds_bytes = std::max((size_t)staged.num_rows * rows * RED_THREADS * 4, sizeof(EF) * RED_THREADS);
        {
            const uint32_t grid = whole_domains(round0_cached_grid(staged, log_lifted));
#define LAUNCH(R, NOUT)                                                                                             \
    OVM_LAUNCH((round0_cached_kernel<R, NOUT>), dim3(grid), dim3(RED_THREADS), lds_bytes, s, staged, lde, \
                       n_cols_total, log_lifted, l_skip, n_cosets, n_out, d_eq_xi, add_const, ro.partials)
            if (n_out == 1) {
                if (rows == 4) LAUNCH(4, 1);
                else if (rows == 2) LAUNCH(2, 1);
                else LAUNCH(1, 1);
            } else {
                if (rows == 2) LAUNCH(2, 2);
                else LAUNCH(1, 2);
            }
#undef LAUNCH
            OVM_CHECK_LAUNCH();
            return sum_partials(ro, grid, n_cosets * n_out << l_skip, s);
        }
    }
    uint64_t lanes = (1ull << log_lifted) / rows;
    if (lanes < RED_THREADS) lanes = RED_THREADS;
    uint32_t grid = whole_domains(red_grid(lanes));
#define LAUNCH(NS, R, NOUT)                                                                                      \
    OVM_LAUNCH((round0_eval_kernel<NS, R, NOUT>), dim3(grid), dim3(RED_THREADS), 0, s, prog, lde,       \
                       n_cols_total, log_lifted, l_skip, n_cosets, n_out, d_eq_xi, add_const, ro.partials)
#define CALL(NS)                                           \
    do {                                                   \
        if (n_out == 1) {                                  \
            if (rows == 4) LAUNCH(NS, 4, 1);               \
            else if (rows == 2) LAUNCH(NS, 2, 1);          \
            else LAUNCH(NS, 1, 1);                         \
        } else {                                           \
            if (rows == 2) LAUNCH(NS, 2, 2);               \
            else LAUNCH(NS, 1, 2);                         \
        }                                                  \
    } while (0)
    DISPATCH_SLOTS(prog.num_slots, CALL);
#undef CALL
#undef LAUNCH
    OVM_CHECK_LAUNCH();
    return sum_partials(ro, grid, n_cosets * n_out << l_skip, s);
}

// ---- fold_ple: evaluate each chunk's interpolant at r via barycentric weights ---------------
// twins (optional): twins[j] = column index of the rotated view of column j when this lane should
// produce it as well (same 17 cells, weights shifted by one row), -2 = column j is such a rotated view
// and is skipped here, -1 = neither.
// (The chunks as whole 1 KiB runs through LDS, wave_chunks.cuh, did nothing here: 2.95 -> 3.19 and 3.00 -> 2.96 ms for the
// two launches of config B -- the 16-byte stores of the folded values and the weights are this kernel's other half.)
__global__ void fold_ple_kernel(const ColDesc *cols, const int32_t *twins, uint32_t log_lifted, uint32_t l_skip,
                                const EF *weights, EF *out) {
    const uint64_t nx = 1ull << (log_lifted - l_skip);
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nx) return;
    const int32_t twin = twins ? twins[blockIdx.y] : -1;
    if (twin == -2) return;
    const ColDesc cd = cols[blockIdx.y];
    const uint64_t hmask = (1ull << cd.log_height) - 1;
    EF acc = EF::zero();
    if (l_skip == 4 && cd.log_height >= 4) {
        // 16-row chunks: the lane's chunk is 64 contiguous bytes -> four 16-byte loads (a rotated
        // view needs the first row of the next chunk as well)
        const uint64_t base = (x << 4) & hmask;
        const uint4 *src = reinterpret_cast<const uint4 *>(cd.ptr + base);
        uint32_t r[17];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 t = src[q];
            r[4 * q] = t.x;
            r[4 * q + 1] = t.y;
            r[4 * q + 2] = t.z;
            r[4 * q + 3] = t.w;
        }
        r[16] = (cd.rot || twin >= 0) ? cd.ptr[(base + 16) & hmask].v : 0;
        if (cd.rot > 1) {  // (views only ever rotate by 0 or 1; keep the general path exact)
            for (uint32_t z = 0; z < 16; z++) r[z] = cd.ptr[((x << 4) + z + cd.rot) & hmask].v;
            r[16] = 0;
        }
        const uint32_t sh = cd.rot == 1 ? 1 : 0;
        Acc96 lazy;
        lazy.clear();
#pragma unroll
        for (int z = 0; z < 16; z++) mac_weight(lazy, weights, z, sh ? r[z + 1] : r[z]);
        acc = lazy.finish();
        if (twin >= 0 && cd.rot == 0) {  // the rotated view of the same cells
            Acc96 lazy1;
            lazy1.clear();
#pragma unroll
            for (int z = 0; z < 16; z++) mac_weight(lazy1, weights, z, r[z + 1]);
            st_ef(out + (uint64_t)twin * nx + x, lazy1.finish());
        }
    } else {
        for (uint32_t z = 0; z < (1u << l_skip); z++) {
            Fp v = cd.ptr[((x << l_skip) + z + cd.rot) & hmask];
            acc = acc + ld_ef(weights + z) * v;
        }
    }
    st_ef(out + (uint64_t)blockIdx.y * nx + x, acc);
}
int fold_ple(const ColDesc *d_cols, const int32_t *d_twins, uint32_t n_cols, uint32_t log_lifted, uint32_t l_skip,
             const EF *d_weights, EF *out, hipStream_t s) {
    if (n_cols == 0) return 0;
    if (d_twins && l_skip != 4) d_twins = nullptr;  // (only the 16-row chunk path shares the cells)
    uint64_t nx = 1ull << (log_lifted - l_skip);
    OVM_LAUNCH(fold_ple_kernel, dim3(ceil_div(nx, 256), n_cols), dim3(256), 0, s, d_cols, d_twins, log_lifted, l_skip,
                       d_weights, out);
    OVM_CHECK_LAUNCH();
    return 0;
}

// ---- multilinear rounds ----------------------------------------------------------------------
struct MleFetch {
    const EF *mats;
    uint32_t log_h;
    uint64_t y;
    Fp x;
    __device__ EF operator()(uint32_t col) const {
        const EF *p = mats + ((uint64_t)col << log_h) + 2 * y;
        EF t0 = ld_ef(p), t1 = ld_ef(p + 1);
        return t0 + (t1 - t0) * x;
    }
};

// ---- the same rounds with all D evaluation points in one lane ----------------------------------
// A lane owns one pair y and evaluates the program at X = 1..D together: one instruction decode and
// ONE fetch of the pair (t0, t1) per column operand serve all points -- the values at X = 1, 2, ...
// are t1, t1 + d, t1 + 2d with d = t1 - t0, i.e. additions only, where the per-X kernels multiply.
template <int NS, int D>
__device__ __forceinline__ void mle_eval_points(const Instr *code, uint32_t n_instr, const EF *weights, const EF *mats,
                                                uint32_t log_h, uint64_t y, EF (&out)[D][2]) {
    EF slots[NS][D];
    auto operand = [&](uint8_t kind, uint32_t payload, EF (&v)[D]) {
        if (kind == K_SLOT) {
#pragma unroll
            for (int x = 0; x < D; x++) v[x] = slots[payload][x];
        } else if (kind == K_COL) {
            const EF *p = mats + ((uint64_t)payload << log_h) + 2 * y;
            const EF t0 = ld_ef(p), t1 = ld_ef(p + 1);
            const EF d = t1 - t0;
            v[0] = t1;
#pragma unroll
            for (int x = 1; x < D; x++) v[x] = v[x - 1] + d;
        } else {
            const EF cst = EF::from_base(Fp::raw(payload));
#pragma unroll
            for (int x = 0; x < D; x++) v[x] = cst;
        }
    };
    for (uint32_t pc = 0; pc < n_instr; pc++) {
        const Instr in = code[pc];
        EF a[D], b[D];
        operand(in.ak, in.a, a);
        switch (in.op) {
            case OP_ADD:
                operand(in.bk, in.b, b);
#pragma unroll
                for (int x = 0; x < D; x++) slots[in.dst][x] = a[x] + b[x];
                break;
            case OP_SUB:
                operand(in.bk, in.b, b);
#pragma unroll
                for (int x = 0; x < D; x++) slots[in.dst][x] = a[x] - b[x];
                break;
            case OP_MUL:
                operand(in.bk, in.b, b);
#pragma unroll
                for (int x = 0; x < D; x++) slots[in.dst][x] = a[x] * b[x];
                break;
            case OP_NEG:
#pragma unroll
                for (int x = 0; x < D; x++) slots[in.dst][x] = -a[x];
                break;
            default: {  // OP_ACC
                const EF::Pre w = ld_ef(weights + in.b).pre();
                if (in.acc) {
#pragma unroll
                    for (int x = 0; x < D; x++) out[x][1] = out[x][1] + a[x] * w;
                } else {
#pragma unroll
                    for (int x = 0; x < D; x++) out[x][0] = out[x][0] + a[x] * w;
                }
                break;
            }
        }
    }
}

template <int NS, int D>
__device__ __forceinline__ void mle_round_points_body(const Instr *gcode, uint32_t n_instr, const EF *weights,
                                                      const EF *mats, uint32_t log_h, uint32_t n_out, const EF *eq,
                                                      EF add_const, uint32_t gx, EF (&sums)[2 * D]) {
    const uint64_t n_y = 1ull << (log_h - 1);
    const uint64_t stride = (uint64_t)gx * RED_THREADS;
#pragma unroll
    for (int k = 0; k < 2 * D; k++) sums[k] = EF::zero();
    for (uint64_t y = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; y < n_y; y += stride) {
        const EF::Pre e = ld_ef(eq + y).pre();
        EF out[D][2];
#pragma unroll
        for (int x = 0; x < D; x++) {
            out[x][0] = EF::zero();
            out[x][1] = add_const;
        }
        mle_eval_points<NS, D>(gcode, n_instr, weights, mats, log_h, y, out);
#pragma unroll
        for (int x = 0; x < D; x++) {
            sums[2 * x] = sums[2 * x] + out[x][0] * e;
            if (n_out > 1) sums[2 * x + 1] = sums[2 * x + 1] + out[x][1] * e;
        }
    }
}

template <int NS, int D>
__global__ __launch_bounds__(RED_THREADS) void mle_round_points_kernel(const Instr *code, const uint32_t *off,
                                                                       const EF *weights, const EF *mats,
                                                                       uint32_t log_h, uint32_t n_out, const EF *eq,
                                                                       EF add_const, RedOut ro) {
    const uint32_t grp = blockIdx.z;
    EF sums[2 * D];
    mle_round_points_body<NS, D>(code + off[grp], off[grp + 1] - off[grp], weights, mats, log_h, n_out, eq,
                                 grp == 0 ? add_const : EF::zero(), gridDim.x, sums);
    block_sum_out<2 * D>(sums, ro, ((size_t)(blockIdx.x * gridDim.z + grp)) * (2 * D));
    finish_reduction<8>(ro, gridDim.x * gridDim.z, 2 * D);
}

// ---- the same with the slots in LDS: the word executor of mle_vm.cuh ---------------------------------------------------
// mle_eval_points keeps EF slots[NS][D] in a dynamically indexed array, which the compiler places in scratch memory:
// every interpreted instruction then reads up to 2 x D x 16 bytes of scratch and writes D x 16 (400 B of private segment,
// 150 VGPRs), and a column operand is a dependent global load issued where it is used. On the real Keccak-f AIR
// (42 514 instructions, 2^20 rows: config B) that was ~600 GB of scratch traffic per proof and 113 ms of multilinear
// rounds. Here the program is in the accumulator form of program.h (MleWord) and runs on the assembly executor
// MleVm<D>::run: slots are LDS rows of D x 4 KiB per workgroup (programs of up to MLE_LDS_MAX_ROWS / D stored slots
// qualify -- the demand-driven compiler and compact_slots get the real programs there), three workgroups per CU up to
// D = 3.
static constexpr uint32_t MLE_LDS_MAX_ROWS = 19;  // stored slots x D (KiB of LDS per workgroup / 4)
template <int D>
__global__ __launch_bounds__(RED_THREADS, D <= 3 ? 3 : 2) void mle_round_points_lds_kernel(
    const MleWord *__restrict__ words, const uint32_t *__restrict__ woff, const MleWeight *__restrict__ weights,
    const EF *__restrict__ mats, uint32_t log_h, uint32_t n_out, const EF *__restrict__ eq, EF add_const, RedOut ro) {
    extern __shared__ uint4 mle_lds[];
    const uint32_t grp = blockIdx.z;
    EF sums[2 * D];
    MleVm<D>::run(words + woff[grp], woff[grp + 1] - woff[grp] - 2 * MLE_WORD_GROUP, weights, mats, log_h, n_out, eq,
                  grp == 0 ? add_const : EF::zero(), blockIdx.x, gridDim.x, mle_lds, sums);
    block_sum_out<2 * D>(sums, ro, ((size_t)(blockIdx.x * gridDim.z + grp)) * (2 * D));
    finish_reduction<8>(ro, gridDim.x * gridDim.z, 2 * D);
}

uint32_t mle_round_gx(uint32_t log_h, uint32_t /*degree*/, uint32_t n_groups) {
    uint32_t gx = red_grid(1ull << (log_h - 1));
    // keep the total grid near RED_BLOCKS workgroups
    uint32_t cap = RED_BLOCKS / n_groups;
    if (cap < 1) cap = 1;
    return gx > cap ? cap : gx;
}

// Tail of the batched kernels: the workgroup's 2 D sums become its record; the job's last workgroup sums the job's
// records into the job's mailbox slot; the last job raises the flag.
template <int D>
__device__ __forceinline__ void mle_job_finish(EF (&sums)[2 * D], const MleJob &job, uint32_t j, uint32_t grp, uint32_t n_jobs,
                                               uint32_t *counters, const RedOut &ro) {
    constexpr uint32_t K = 2 * D;
    block_sum_store<2 * D>(sums, job.partials + ((size_t)(blockIdx.x * job.n_groups + grp)) * K, true);
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) s_last = atomicAdd(counters + 1 + j, 1u) == job.gx * job.n_groups - 1;
    __syncthreads();
    if (!s_last) return;
    acquire_from_device();
    EF acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = EF::zero();
    const uint32_t n_records = job.gx * job.n_groups;
    for (uint32_t b = threadIdx.x; b < n_records; b += RED_THREADS) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k < K) acc[k] = acc[k] + ld_ef_agent(job.partials + (size_t)b * K + k);
    }
    block_sum_store<8, true>(acc, job.out);  // (released to the host)
    if (threadIdx.x == 0) {
        counters[1 + j] = 0;
        if (atomicAdd(counters, 1u) == n_jobs - 1) {
            counters[0] = 0;
            publish(ro);
        }
    }
}

// Batched short rounds: lanes enumerate (pair y, point X) of a (job, group), so a round of 8 pairs and
// 4 points fills 32 lanes of ONE wave instead of 8 lanes of four (the short rounds are bound by the
// number of mostly-empty waves the machine has to run, not by latency: prefetching operands or
// splitting the groups finer both made them slower).
template <int NS, int D>
__global__ __launch_bounds__(RED_THREADS) void mle_round_batched_kernel(const MleJob *jobs, uint32_t n_jobs,
                                                                        uint32_t *counters, RedOut ro) {
    uint32_t j = 0;
    j = find_job(n_jobs, blockIdx.z, [&](uint32_t i) { return jobs[i].z0; });
    const MleJob job = jobs[j];
    if (blockIdx.x >= job.gx) return;
    const uint64_t n_items = (uint64_t)D << (job.log_h - 1);  // (y, X) pairs
    const uint64_t stride = (uint64_t)job.gx * RED_THREADS;
    const uint32_t grp = blockIdx.z - job.z0, K = 2 * D;
    const Instr *gcode = job.code + job.off[grp];
    const uint32_t n_instr = job.off[grp + 1] - job.off[grp];
    EF sums[2 * D];
#pragma unroll
    for (int k = 0; k < 2 * D; k++) sums[k] = EF::zero();
    EF slots[NS];
    for (uint64_t item = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; item < n_items; item += stride) {
        const uint64_t y = item / D;
        const uint32_t xi = (uint32_t)(item % D);  // point X = xi + 1
        EF e = ld_ef(job.eq + y);
        MleFetch fetch{job.mats, job.log_h, y, Fp::from_canonical(xi + 1)};
        EF out[2] = {EF::zero(), grp == 0 ? job.add_const : EF::zero()};
        run_program_acc<EF>(gcode, n_instr, job.weights, slots, fetch, out);
        const EF s0 = e * out[0], s1 = job.n_out > 1 ? e * out[1] : EF::zero();
#pragma unroll
        for (int x = 0; x < D; x++)
            if (xi == (uint32_t)x) {
                sums[2 * x] = sums[2 * x] + s0;
                sums[2 * x + 1] = sums[2 * x + 1] + s1;
            }
    }
    mle_job_finish<D>(sums, job, j, grp, n_jobs, counters, ro);
}

// The same batch on the accumulator-form executor of the large rounds (mle_vm.cuh: MleVm<D>::run): for jobs of at
// least a workgroup's worth of pairs. A many-AIR proof has dozens of wide, short traces with programs of 1 000-10 000
// instructions (a reth segment: 46 AIRs, 12 815 columns): in the kernel above every column operand of theirs is a dependent
// global load in front of its instruction and four lanes fetch the same pair, and that kernel was the largest of the
// multilinear rounds (9.6 of 23 ms); here the operands of four instructions are in flight together and a pair is
// fetched once.
template <int D>
__global__ __launch_bounds__(RED_THREADS, D <= 3 ? 3 : 2) void mle_round_batched_lds_kernel(const MleJob *__restrict__ jobs,
                                                                                          uint32_t n_jobs, uint32_t *counters,
                                                                                          RedOut ro) {
    extern __shared__ uint4 mle_lds[];
    const uint32_t j = find_job(n_jobs, blockIdx.z, [&](uint32_t i) { return jobs[i].z0; });
    const MleJob job = jobs[j];
    if (blockIdx.x >= job.gx) return;
    const uint32_t grp = blockIdx.z - job.z0;
    EF sums[2 * D];
    MleVm<D>::run(job.words + job.woff[grp], job.woff[grp + 1] - job.woff[grp] - 2 * MLE_WORD_GROUP, job.weights_pre, job.mats,
                  job.log_h, job.n_out, job.eq, grp == 0 ? job.add_const : EF::zero(), blockIdx.x, job.gx, mle_lds, sums);
    mle_job_finish<D>(sums, job, j, grp, n_jobs, counters, ro);
}

// workgroups per (job, group) of the batched kernel
uint32_t mle_round_batched_gx(uint32_t log_h, uint32_t degree, uint32_t n_groups) {
    uint32_t gx = red_grid((uint64_t)degree << (log_h - 1));
    uint32_t cap = RED_BLOCKS / n_groups;
    if (cap < 1) cap = 1;
    return gx > cap ? cap : gx;
}

int mle_round_eval_batched(const MleJob *d_jobs, const MleJob *h_jobs, uint32_t n_jobs, uint32_t degree,
                           uint32_t *d_counters, const RedOut &ro, hipStream_t s) {
    if (n_jobs == 0) return 0;
    if (degree == 0 || degree > 4) return (int)hipErrorInvalidValue;
    uint32_t gx = 1, gz = 0, max_slots = 1;
    for (uint32_t j = 0; j < n_jobs; j++) {
        if (h_jobs[j].log_h == 0 || h_jobs[j].n_groups == 0 || h_jobs[j].n_out > 2 || h_jobs[j].z0 != gz ||
            (uint64_t)h_jobs[j].gx * h_jobs[j].n_groups * 2 * degree > h_jobs[j].partials_cap)
            return (int)hipErrorInvalidValue;
        gx = std::max(gx, h_jobs[j].gx);
        gz += h_jobs[j].n_groups;
        max_slots = std::max(max_slots, h_jobs[j].max_slots);
    }
    // (all points of a pair in ONE lane, as the large rounds do, was measured slower here: 4.8 -> 6.8 ms on
    // the config B shape)
    dim3 grid(gx, 1, gz);
#define CALLD(NS, DD) \
    OVM_LAUNCH((mle_round_batched_kernel<NS, DD>), grid, dim3(RED_THREADS), 0, s, d_jobs, n_jobs, d_counters, ro)
#define CALL(NS)                                  \
    do {                                          \
        if (degree == 1) CALLD(NS, 1);            \
        else if (degree == 2) CALLD(NS, 2);       \
        else if (degree == 3) CALLD(NS, 3);       \
        else CALLD(NS, 4);                        \
    } while (0)
    DISPATCH_SLOTS(max_slots, CALL);
#undef CALL
#undef CALLD
    OVM_CHECK_LAUNCH();
    return 0;
}

int mle_round_eval_batched_lds(const MleJob *d_jobs, const MleJob *h_jobs, uint32_t n_jobs, uint32_t degree,
                               uint32_t *d_counters, const RedOut &ro, hipStream_t s) {
    if (n_jobs == 0) return 0;
    if (degree == 0 || degree > 4) return (int)hipErrorInvalidValue;
    uint32_t gx = 1, gz = 0, max_slots = 1;
    for (uint32_t j = 0; j < n_jobs; j++) {
        const uint32_t rows = h_jobs[j].lds_slots;  // (of the words: MleWords::rows)
        if (h_jobs[j].log_h == 0 || h_jobs[j].n_groups == 0 || h_jobs[j].n_out > 2 || h_jobs[j].z0 != gz ||
            rows * degree > MLE_LDS_MAX_ROWS || !h_jobs[j].words || !h_jobs[j].woff || !h_jobs[j].weights_pre ||
            (uint64_t)h_jobs[j].gx * h_jobs[j].n_groups * 2 * degree > h_jobs[j].partials_cap)
            return (int)hipErrorInvalidValue;
        gx = std::max(gx, h_jobs[j].gx);
        gz += h_jobs[j].n_groups;
        max_slots = std::max(max_slots, rows);
    }
    dim3 grid(gx, 1, gz);
    const size_t lds_bytes = (size_t)max_slots * degree * 4 * RED_THREADS * sizeof(uint32_t);
#define CALLD(DD) \
    OVM_LAUNCH((mle_round_batched_lds_kernel<DD>), grid, dim3(RED_THREADS), lds_bytes, s, d_jobs, n_jobs, d_counters, ro)
    if (degree == 1) CALLD(1);
    else if (degree == 2) CALLD(2);
    else if (degree == 3) CALLD(3);
    else CALLD(4);
#undef CALLD
    OVM_CHECK_LAUNCH();
    return 0;
}
bool mle_lds_eligible(uint32_t lds_slots, uint32_t degree) {
    return tuning().mle_lds_slots && lds_slots * degree <= MLE_LDS_MAX_ROWS;
}

int mle_round_eval_grouped(const Instr *d_code, const uint32_t *d_off, uint32_t n_groups, uint32_t max_slots, const MleWord *d_words,
                           const uint32_t *d_woff, uint32_t lds_rows, const EF *d_weights, const MleWeight *d_weights_pre,
                           const EF *mats, uint32_t log_h, uint32_t degree, uint32_t n_out, const EF *d_eq, EF add_const,
                           const RedOut &ro, hipStream_t s) {
    if (degree == 0 || degree > 4 || n_out > 2 || log_h == 0 || n_groups == 0) return (int)hipErrorInvalidValue;
    const uint32_t gx = mle_round_gx(log_h, degree, n_groups);
    if (d_words && mle_lds_eligible(lds_rows, degree)) {
        dim3 grid(gx, 1, n_groups);
        const size_t lds_bytes = (size_t)lds_rows * degree * 4 * RED_THREADS * sizeof(uint32_t);
#define CALLD(DD)                                                                                                         \
    OVM_LAUNCH((mle_round_points_lds_kernel<DD>), grid, dim3(RED_THREADS), lds_bytes, s, d_words, d_woff, d_weights_pre, mats, \
                       log_h, n_out, d_eq, add_const, red_inline(ro, gx * n_groups, 2 * degree))
        if (degree == 1) CALLD(1);
        else if (degree == 2) CALLD(2);
        else if (degree == 3) CALLD(3);
        else CALLD(4);
#undef CALLD
        OVM_CHECK_LAUNCH();
        return red_finish(ro, gx * n_groups, gx * n_groups, 2 * degree, s);
    }
    // programs whose slots do not fit the LDS image: slots in private memory
    dim3 grid(gx, 1, n_groups);
#define CALLD(NS, DD)                                                                                                  \
    OVM_LAUNCH((mle_round_points_kernel<NS, DD>), grid, dim3(RED_THREADS), 0, s, d_code, d_off, d_weights, mats, \
                       log_h, n_out, d_eq, add_const, red_inline(ro, gx * n_groups, 2 * degree))
#define CALL(NS)                                  \
    do {                                          \
        if (degree == 1) CALLD(NS, 1);            \
        else if (degree == 2) CALLD(NS, 2);       \
        else if (degree == 3) CALLD(NS, 3);       \
        else CALLD(NS, 4);                        \
    } while (0)
        DISPATCH_SLOTS(max_slots, CALL);
#undef CALL
#undef CALLD
    OVM_CHECK_LAUNCH();
    return red_finish(ro, gx * n_groups, gx * n_groups, 2 * degree, s);
}

__global__ void fold_mle_kernel(const EF *in, EF *out, uint32_t log_h, EF r_in) {
    const EF::Pre r = r_in.pre();
    const uint64_t h2 = 1ull << (log_h - 1);
    uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= h2) return;
    const EF *p = in + ((uint64_t)blockIdx.y << log_h) + 2 * y;
    EF t0 = ld_ef(p), t1 = ld_ef(p + 1);
    st_ef(out + (uint64_t)blockIdx.y * h2 + y, t0 + (t1 - t0) * r);
}
int fold_mle(const EF *in, EF *out, uint32_t n_cols, uint32_t log_h, EF r, hipStream_t s) {
    if (n_cols == 0 || log_h == 0) return 0;
    OVM_LAUNCH(fold_mle_kernel, dim3(ceil_div(1ull << (log_h - 1), 256), n_cols), dim3(256), 0, s, in, out,
                       log_h, r);
    OVM_CHECK_LAUNCH();
    return 0;
}

__global__ void fold_mle_batched_kernel(const FoldJob *jobs, uint32_t n_jobs, EF r_in) {
    uint32_t j = 0;
    j = find_job(n_jobs, blockIdx.y, [&](uint32_t i) { return jobs[i].y0; });
    const FoldJob job = jobs[j];
    const EF::Pre r = r_in.pre();
    const uint64_t h2 = 1ull << (job.log_h - 1);
    const uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= h2) return;
    const uint32_t col = blockIdx.y - job.y0;
    const EF *p = job.in + ((uint64_t)col << job.log_h) + 2 * y;
    EF t0 = ld_ef(p), t1 = ld_ef(p + 1);
    st_ef(job.out + (uint64_t)col * h2 + y, t0 + (t1 - t0) * r);
}
// up to FoldJobsArg::MAX jobs travel as a kernel argument (no upload in front of every round)
__global__ void fold_mle_batched_arg_kernel(FoldJobsArg a, uint32_t n_jobs, EF r_in) {
    uint32_t j = 0;
    j = find_job(n_jobs, blockIdx.y, [&](uint32_t i) { return a.j[i].y0; });
    const FoldJob job = a.j[j];
    const EF::Pre r = r_in.pre();
    const uint64_t h2 = 1ull << (job.log_h - 1);
    const uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= h2) return;
    const uint32_t col = blockIdx.y - job.y0;
    const EF *p = job.in + ((uint64_t)col << job.log_h) + 2 * y;
    EF t0 = ld_ef(p), t1 = ld_ef(p + 1);
    st_ef(job.out + (uint64_t)col * h2 + y, t0 + (t1 - t0) * r);
}
int fold_mle_batched(const FoldJob *d_jobs, const FoldJob *h_jobs, uint32_t n_jobs, EF r, hipStream_t s) {
    if (n_jobs == 0) return 0;
    uint64_t gx = 1, gy = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        if (h_jobs[j].log_h == 0 || h_jobs[j].n_cols == 0 || h_jobs[j].y0 != gy) return (int)hipErrorInvalidValue;
        gx = std::max<uint64_t>(gx, ceil_div(1ull << (h_jobs[j].log_h - 1), 256));
        gy += h_jobs[j].n_cols;
    }
    if (gy > 65535) return (int)hipErrorInvalidValue;
    if (!d_jobs) {
        if (n_jobs > FoldJobsArg::MAX) return (int)hipErrorInvalidValue;
        FoldJobsArg a;
        for (uint32_t j = 0; j < n_jobs; j++) a.j[j] = h_jobs[j];
        OVM_LAUNCH(fold_mle_batched_arg_kernel, dim3((uint32_t)gx, (uint32_t)gy), dim3(256), 0, s, a, n_jobs, r);
        OVM_CHECK_LAUNCH();
        return 0;
    }
    OVM_LAUNCH(fold_mle_batched_kernel, dim3((uint32_t)gx, (uint32_t)gy), dim3(256), 0, s, d_jobs, n_jobs, r);
    OVM_CHECK_LAUNCH();
    return 0;
}

// ---- LogUp GKR -------------------------------------------------------------------------------
// All GKR data (the fraction tree, the fold buffers, the eq tables) is LIMB-PLANAR: limb k of
// element i lives at base[k * stride + i]. A lane that needs the 2, 4 or 8 consecutive elements
// of its pair therefore issues one dwordx2/x4 load per limb plane and consecutive lanes touch
// consecutive addresses -- every load and store instruction of every sub-round is unit-stride
// across the wavefront, at any fold depth. (With 16-byte AoS elements a lane streamed 256 B of
// its own and the PMC counters showed 4-5x the algorithmic HBM traffic; profiles/r01.)
//
// "Layer format": an array of 2m fractions; record x (the sumcheck's evaluation point x) is
// (p0, q0, p1, q1) = (P[2x], Q[2x], P[2x+1], Q[2x+1]). A tree layer is in this format by
// construction, and the folded buffers keep it.
__device__ __forceinline__ EF ldp(const uint32_t *b, uint64_t st, uint64_t i) {
    EF r;
#pragma unroll
    for (int k = 0; k < 4; k++) r.c[k] = Fp::raw(b[k * st + i]);
    return r;
}
__device__ __forceinline__ void ldp2(const uint32_t *b, uint64_t st, uint64_t i, EF &e0, EF &e1) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint2 v = *reinterpret_cast<const uint2 *>(b + k * st + i);
        e0.c[k] = Fp::raw(v.x);
        e1.c[k] = Fp::raw(v.y);
    }
}
__device__ __forceinline__ void stp2(uint32_t *b, uint64_t st, uint64_t i, const EF &e0, const EF &e1) {
#pragma unroll
    for (int k = 0; k < 4; k++) *reinterpret_cast<uint2 *>(b + k * st + i) = make_uint2(e0.c[k].v, e1.c[k].v);
}
__device__ __forceinline__ void ldp4(const uint32_t *b, uint64_t st, uint64_t i, EF (&e)[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 v = *reinterpret_cast<const uint4 *>(b + k * st + i);
        e[0].c[k] = Fp::raw(v.x);
        e[1].c[k] = Fp::raw(v.y);
        e[2].c[k] = Fp::raw(v.z);
        e[3].c[k] = Fp::raw(v.w);
    }
}
__device__ __forceinline__ void stp4(uint32_t *b, uint64_t st, uint64_t i, const EF (&e)[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++)
        *reinterpret_cast<uint4 *>(b + k * st + i) = make_uint4(e[0].c[k].v, e[1].c[k].v, e[2].c[k].v, e[3].c[k].v);
}

// Leaf numerators are base-field values (count * norm): limbs 1..3 of the leaf layer's P planes
// are identically zero. With p_base the leaf layer's readers (PBASE kernels below) load limb 0
// only and nobody writes or reads the other three planes -- 3/8 of the leaf layer's traffic.
__device__ __forceinline__ void ldp2_base(const uint32_t *b, uint64_t i, EF &e0, EF &e1) {
    uint2 v = *reinterpret_cast<const uint2 *>(b + i);
    e0 = EF::from_base(Fp::raw(v.x));
    e1 = EF::from_base(Fp::raw(v.y));
}
__device__ __forceinline__ void ldp4_base(const uint32_t *b, uint64_t i, EF (&e)[4]) {
    uint4 v = *reinterpret_cast<const uint4 *>(b + i);
    e[0] = EF::from_base(Fp::raw(v.x));
    e[1] = EF::from_base(Fp::raw(v.y));
    e[2] = EF::from_base(Fp::raw(v.z));
    e[3] = EF::from_base(Fp::raw(v.w));
}

template <int NS>
struct InterpLeaves {
    DevProgram prog;
    const EF *leaf_const;
    __device__ __forceinline__ void operator()(const BaseFetch &fetch, const LeafEmit &emit) const {
        Fp slots[NS];
        run_program_leaves<Fp>(prog.code, prog.n_instr, prog.weights, leaf_const, slots, fetch, emit);
    }
};
template <int NS>
__global__ void gkr_leaves_kernel(DevProgram prog, const EF *leaf_const, const ColDesc *cols, uint32_t log_height,
                                  const uint64_t *leaf_base, const uint32_t *leaf_log_len, Fp norm,
                                  FracPlanes leaves) {
    gkr_leaves_body(InterpLeaves<NS>{prog, leaf_const}, cols, log_height, leaf_base, leaf_log_len, norm, leaves);
}
// Cached-column form (program.h: stage_cached on the DEN/LEAF program; cached_vm.cuh: CachedCore<1, 1, true>): one row per
// lane, trace columns by pointer (the load records were resolved against the trace's column descriptors on the host:
// Prover::make_leaf_prog), a denominator term is an accumulation, a LEAF marker emits.
__device__ __forceinline__ void gkr_leaves_cached_body(const DevCached &prog, const EF *leaf_const, uint32_t log_height,
                                                       const uint64_t *leaf_base, const uint32_t *leaf_log_len, Fp norm,
                                                       FracPlanes leaves, uint32_t bx) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t row = (uint64_t)bx * blockDim.x + threadIdx.x;
    const bool live = row < (1ull << log_height);
    const uint64_t at = live ? row : 0;  // (dead lanes of a short trace read row 0 and emit nothing)
    const uint64_t hmask = (1ull << log_height) - 1;
    const uint32_t o[4] = {(uint32_t)at << 2, (uint32_t)((at + 1) & hmask) << 2, 0, 0};
    const LeafEmit emit{leaves, leaf_base, leaf_log_len, log_height, row, norm};
    Acc96 acc0[1], acc1[1];
    acc0[0].clear(), acc1[0].clear();
    CachedCore<1, 1, true>::run(prog, 0, 0, o, acc0, acc1, [&](uint32_t k, Fp numer, const Acc96 &den_sum) {
        const EF den = den_sum.finish() + ld_ef(leaf_const + k);
        if (live) emit(k, numer, den);
    });
#endif
}
__global__ __launch_bounds__(256) void gkr_leaves_cached_kernel(DevCached prog, const EF *leaf_const, uint32_t log_height,
                                                                const uint64_t *leaf_base, const uint32_t *leaf_log_len,
                                                                Fp norm, FracPlanes leaves) {
    gkr_leaves_cached_body(prog, leaf_const, log_height, leaf_base, leaf_log_len, norm, leaves, blockIdx.x);
}
__global__ __launch_bounds__(256) void gkr_leaves_cached_batched_kernel(const LeavesJob *jobs, uint32_t n_jobs,
                                                                        FracPlanes leaves) {
    uint32_t j = 0;
    j = find_job(n_jobs, blockIdx.x, [&](uint32_t i) { return jobs[i].wg0; });
    const LeavesJob job = jobs[j];
    gkr_leaves_cached_body(job.prog, job.leaf_const, job.log_height, job.leaf_base, job.leaf_log_len, job.norm, leaves,
                           blockIdx.x - job.wg0);
}
int gkr_eval_leaves_batched(const LeavesJob *d_jobs, const LeavesJob *h_jobs, uint32_t n_jobs, FracPlanes leaves,
                            hipStream_t s) {
    if (n_jobs == 0) return 0;
    uint32_t grid = 0;
    size_t lds_bytes = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        if (!h_jobs[j].prog.words || h_jobs[j].prog.rows != 1 || h_jobs[j].wg0 != grid || h_jobs[j].gx == 0)
            return (int)hipErrorInvalidValue;
        grid += h_jobs[j].gx;
        lds_bytes = std::max(lds_bytes, (size_t)h_jobs[j].prog.num_rows * RED_THREADS * 4);
    }
    OVM_LAUNCH(gkr_leaves_cached_batched_kernel, dim3(grid), dim3(256), lds_bytes, s, d_jobs, n_jobs, leaves);
    OVM_CHECK_LAUNCH();
    return 0;
}

int gkr_eval_leaves(const DevProgram &prog, const DevCached &staged, const EF *d_leaf_const, const ColDesc *d_cols,
                    uint32_t log_height, const uint64_t *d_leaf_base, const uint32_t *d_leaf_log_len, Fp norm,
                    FracPlanes leaves, hipStream_t s) {
    uint32_t grid = ceil_div(1ull << log_height, 256);
    if (staged.words) {
        const size_t lds_bytes = (size_t)staged.num_rows * RED_THREADS * 4;
        OVM_LAUNCH(gkr_leaves_cached_kernel, dim3(grid), dim3(256), lds_bytes, s, staged, d_leaf_const, log_height, d_leaf_base,
                           d_leaf_log_len, norm, leaves);
        OVM_CHECK_LAUNCH();
        return 0;
    }
#define CALL(NS)                                                                                           \
    OVM_LAUNCH(gkr_leaves_kernel<NS>, dim3(grid), dim3(256), 0, s, prog, d_leaf_const, d_cols, log_height, \
                       d_leaf_base, d_leaf_log_len, norm, leaves)
    DISPATCH_SLOTS(prog.num_slots, CALL);
#undef CALL
    OVM_CHECK_LAUNCH();
    return 0;
}

// parent[i] = child[2i] + child[2i+1] as fractions
template <bool PBASE>
__global__ void gkr_build_kernel(FracPlanes child, FracPlanes parent, uint64_t n_parent) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parent) return;
    EF p0, p1, q0, q1;
    if (PBASE)
        ldp2_base(child.P(), 2 * i, p0, p1);
    else
        ldp2(child.P(), child.stride, 2 * i, p0, p1);
    ldp2(child.Q(), child.stride, 2 * i, q0, q1);
    stp(parent.P(), parent.stride, i, p0 * q1 + q0 * p1);
    stp(parent.Q(), parent.stride, i, q0 * q1);
}
// Three levels per pass: 1.875 n of traffic per three levels instead of 3 x 1.125 n. Lane i: children 8i..8i+7 -> parents
// 4i..4i+3, grandparents 2i, 2i+1, great-grandparent i. What the range holds outside those aligned blocks goes to extra lanes
// of the same launch: at most two grandparents whose sibling is outside the range (each with its two parents, from four
// children) and at most two lone parents (from two children).
struct BuildEdges {
    FracPlanes pair_child[2], pair_parent[2], pair_grand[2];
    FracPlanes lone_child[2], lone_parent[2];
    uint32_t n_pair, n_lone;
};
template <bool PBASE>
__device__ __forceinline__ void build_lone(const FracPlanes &c, const FracPlanes &p) {
    EF p0, p1, q0, q1;
    if (PBASE)
        ldp2_base(c.P(), 0, p0, p1);
    else
        ldp2(c.P(), c.stride, 0, p0, p1);
    ldp2(c.Q(), c.stride, 0, q0, q1);
    stp(p.P(), p.stride, 0, p0 * q1 + q0 * p1);
    stp(p.Q(), p.stride, 0, q0 * q1);
}
template <bool PBASE>
__global__ void gkr_build3_kernel(FracPlanes child, FracPlanes parent, FracPlanes grand, FracPlanes great, uint64_t n_great,
                                  BuildEdges edges) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_great) {
        const uint64_t e = i - n_great;
        if (e < edges.n_pair) {
            const FracPlanes c = edges.pair_child[e], p = edges.pair_parent[e], g = edges.pair_grand[e];
            EF cp[4], cq[4];
            if (PBASE)
                ldp4_base(c.P(), 0, cp);
            else
                ldp4(c.P(), c.stride, 0, cp);
            ldp4(c.Q(), c.stride, 0, cq);
            const EF pp0 = cp[0] * cq[1] + cq[0] * cp[1], pq0 = cq[0] * cq[1];
            const EF pp1 = cp[2] * cq[3] + cq[2] * cp[3], pq1 = cq[2] * cq[3];
            stp2(p.P(), p.stride, 0, pp0, pp1);
            stp2(p.Q(), p.stride, 0, pq0, pq1);
            stp(g.P(), g.stride, 0, pp0 * pq1 + pq0 * pp1);
            stp(g.Q(), g.stride, 0, pq0 * pq1);
        } else if (e - edges.n_pair < edges.n_lone) {
            const uint32_t k = (uint32_t)(e - edges.n_pair);
            build_lone<PBASE>(edges.lone_child[k], edges.lone_parent[k]);
        }
        return;
    }
    EF pp[4], pq[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        EF cp[4], cq[4];
        if (PBASE)
            ldp4_base(child.P(), 8 * i + 4 * h, cp);
        else
            ldp4(child.P(), child.stride, 8 * i + 4 * h, cp);
        ldp4(child.Q(), child.stride, 8 * i + 4 * h, cq);
        pp[2 * h] = cp[0] * cq[1] + cq[0] * cp[1], pq[2 * h] = cq[0] * cq[1];
        pp[2 * h + 1] = cp[2] * cq[3] + cq[2] * cp[3], pq[2 * h + 1] = cq[2] * cq[3];
    }
    stp4(parent.P(), parent.stride, 4 * i, pp);
    stp4(parent.Q(), parent.stride, 4 * i, pq);
    const EF gp0 = pp[0] * pq[1] + pq[0] * pp[1], gq0 = pq[0] * pq[1];
    const EF gp1 = pp[2] * pq[3] + pq[2] * pp[3], gq1 = pq[2] * pq[3];
    stp2(grand.P(), grand.stride, 2 * i, gp0, gp1);
    stp2(grand.Q(), grand.stride, 2 * i, gq0, gq1);
    stp(great.P(), great.stride, i, gp0 * gq1 + gq0 * gp1);
    stp(great.Q(), great.stride, i, gq0 * gq1);
}
int gkr_build_three_layers(FracPlanes child, FracPlanes parent, FracPlanes grand, FracPlanes great, uint64_t n_great,
                           const FracPlanes *pair_child, const FracPlanes *pair_parent, const FracPlanes *pair_grand, uint32_t n_pair,
                           const FracPlanes *lone_child, const FracPlanes *lone_parent, uint32_t n_lone, bool child_p_base,
                           hipStream_t s) {
    if (n_pair > 2 || n_lone > 2 || n_great + n_pair + n_lone == 0) return (int)hipErrorInvalidValue;
    BuildEdges e{};
    e.n_pair = n_pair, e.n_lone = n_lone;
    for (uint32_t k = 0; k < n_pair; k++) e.pair_child[k] = pair_child[k], e.pair_parent[k] = pair_parent[k], e.pair_grand[k] = pair_grand[k];
    for (uint32_t k = 0; k < n_lone; k++) e.lone_child[k] = lone_child[k], e.lone_parent[k] = lone_parent[k];
    const dim3 grid(ceil_div(n_great + n_pair + n_lone, 256));
    if (child_p_base)
        OVM_LAUNCH(gkr_build3_kernel<true>, grid, dim3(256), 0, s, child, parent, grand, great, n_great, e);
    else
        OVM_LAUNCH(gkr_build3_kernel<false>, grid, dim3(256), 0, s, child, parent, grand, great, n_great, e);
    OVM_CHECK_LAUNCH();
    return 0;
}
int gkr_build_layer(FracPlanes child, FracPlanes parent, uint64_t n_parent, bool child_p_base, hipStream_t s) {
    if (child_p_base)
        OVM_LAUNCH(gkr_build_kernel<true>, dim3(ceil_div(n_parent, 256)), dim3(256), 0, s, child, parent,
                           n_parent);
    else
        OVM_LAUNCH(gkr_build_kernel<false>, dim3(ceil_div(n_parent, 256)), dim3(256), 0, s, child, parent,
                           n_parent);
    OVM_CHECK_LAUNCH();
    return 0;
}

// sums[X-1] += e * (p0 q1 + p1 q0 + lambda q0 q1)(X) for X = 1, 2, 3 where every quantity is the
// line through its values on records a (X = 0) and b (X = 1). The eq polynomial is FACTORED: the
// weight e = eq(xi_rest, y) covers only the variables not yet bound; the factor of the current
// variable, eq1(xi_sr, X), and the product over the bound ones multiply the three sums on the host
// (prover.hip). That makes the eq table half the size, read-only (never folded) and one load per
// pair instead of four loads and two stores.
//
// NP = 2: only X = 1, 2. What is summed is a QUADRATIC in X once the eq factor is split off, and
// the host knows the round's claim s(0) + s(1), so two points determine it (prover.hip); NP = 3
// is the fallback for the (probability ~2^-124) challenges that make that division impossible.
template <int NP>
__device__ __forceinline__ void gkr_pair_sums(EF (&a)[4], const EF (&b)[4], const EF &e_in, const EF::Pre &lambda,
                                              EF (&sums)[NP]) {
    const EF::Pre e = e_in.pre();
    // p0 q1 + p1 q0 + lambda q0 q1 = p0 q1 + q0 (p1 + lambda q1); t = p1 + lambda q1 is again a
    // line in X, so lambda costs two multiplies per pair instead of one per point: 11 extension
    // multiplies per pair (these kernels are multiply-bound, not bandwidth-bound).
    const EF d0 = b[0] - a[0], d1 = b[1] - a[1], d3 = b[3] - a[3];
    EF t = a[2] + a[3] * lambda;
    const EF td = (b[2] - a[2]) + d3 * lambda;
#pragma unroll
    for (int xi = 0; xi < NP; xi++) {
        a[0] = a[0] + d0;
        a[1] = a[1] + d1;
        a[3] = a[3] + d3;
        t = t + td;
        sums[xi] = sums[xi] + (a[0] * a[3] + a[1] * t) * e;
    }
}

// Heap of partial eq tables for one layer: node 1 = 1; node i at depth k has children
// H[2i] = H[i] * (1 - x), H[2i+1] = H[i] * x with x = point[n-1-k]. Depth k (nodes 2^k .. 2^(k+1))
// is the table over the LAST k variables, low bit = variable n-k: exactly the weights sub-round
// n-1-k needs, so one build serves every sub-round of the layer.
static constexpr uint32_t EQ_HEAP_DIRECT_DEPTH = 10;
// (the point travels as a kernel argument: no upload, no copy command in front of every layer)
__global__ void eq_heap_top_kernel(EqPoint pt, uint32_t n, uint32_t max_depth, EFPlanes heap) {
    __shared__ EF point[EqPoint::MAX];
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) point[j] = pt.x[j];
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // heap node
    if (i == 0 || i >= (2u << max_depth)) return;
    const uint32_t k = 31 - __clz(i), y = i - (1u << k);
    EF acc = EF::one();
    for (uint32_t j = 0; j < k; j++) {
        EF x = point[n - k + j];
        acc = acc * ((y >> j & 1) ? x : EF::one() - x);
    }
    stp(heap.p, heap.stride, i, acc);
}
// ---- windowed eq heap (sharded GKR) ----------------------------------------------------------
// A rank that owns the index window [lo, hi) of a layer's table needs, at depth d of the heap, only
// the entries of its window at that depth (and the parents they are expanded from). Depths up to
// `top` are full tables at heap index 2^d + y as above; deeper depths are stored as windows:
// entry y of depth d lives at heap index off[d] + (y - lo[d]).
EqHeapPlan eq_heap_plan(uint32_t n, const std::vector<std::pair<uint64_t, uint64_t>> &need) {
    EqHeapPlan pl;
    pl.n = n;
    pl.off.assign(n, 0);
    pl.lo.assign(n, 0);
    pl.cnt.assign(n, 0);
    pl.top = n == 0 ? 0 : (n - 1 < EQ_HEAP_DIRECT_DEPTH ? n - 1 : EQ_HEAP_DIRECT_DEPTH);
    if (n == 0) return pl;
    // E[d]: entries depth d must hold = its own window and the parents of everything deeper
    std::vector<std::pair<uint64_t, uint64_t>> E(n, {0, 0});
    bool any = false;
    for (uint32_t d = n; d-- > 0;) {
        uint64_t lo = need[d].first, hi = need[d].second;
        bool has = lo < hi;
        if (d + 1 < n && E[d + 1].first < E[d + 1].second) {
            const uint64_t plo = E[d + 1].first >> 1, phi = ((E[d + 1].second - 1) >> 1) + 1;
            lo = has ? std::min(lo, plo) : plo;
            hi = has ? std::max(hi, phi) : phi;
            has = true;
        }
        E[d] = has ? std::make_pair(lo, hi) : std::make_pair<uint64_t, uint64_t>(0, 0);
        any |= has;
    }
    for (uint32_t d = 0; d <= pl.top; d++) {
        pl.off[d] = uint64_t(1) << d;
        pl.lo[d] = 0;
        pl.cnt[d] = uint64_t(1) << d;
    }
    pl.total = uint64_t(2) << pl.top;
    pl.empty = !any;
    if (!any) return pl;
    for (uint32_t k = pl.top; k + 1 < n;) {
        const uint32_t levels = n - 1 - k < 3 ? n - 1 - k : 3;
        // source lanes: the parents (levels up) of what the deepest level of this launch must hold
        uint64_t lo = E[k + levels].first >> levels, hi = ((E[k + levels].second - 1) >> levels) + 1;
        for (uint32_t l = 1; l < levels; l++) {  // (the intermediate depths' own windows are inside by construction)
            lo = std::min(lo, E[k + l].first >> l);
            hi = std::max(hi, ((E[k + l].second - 1) >> l) + 1);
        }
        lo &= ~uint64_t(7);
        hi = std::min<uint64_t>((hi + 7) & ~uint64_t(7), uint64_t(1) << k);
        EqHeapPlan::Launch ln{k, levels, lo, hi - lo};
        for (uint32_t l = 1; l <= levels; l++) {
            pl.off[k + l] = pl.total;
            pl.lo[k + l] = lo << l;
            pl.cnt[k + l] = (hi - lo) << l;
            pl.total += (pl.cnt[k + l] + 7) & ~uint64_t(7);
        }
        pl.launches.push_back(ln);
        k += levels;
    }
    return pl;
}
// lane t expands entry src_lo + t of depth k into its descendants of depths k+1 .. k+levels
__global__ void eq_heap_levels_win_kernel(EqPoint pt, uint32_t n, uint32_t k, uint32_t levels, EFPlanes heap,
                                          uint64_t src_off, uint64_t src_lo, uint64_t src_first, uint64_t count,
                                          uint64_t off1, uint64_t off2, uint64_t off3, uint64_t dst_lo) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint64_t i = src_first + t;  // entry index at depth k
    EF cur[8];
    cur[0] = ldp(heap.p, heap.stride, src_off + (i - src_lo));
    const uint64_t offs[3] = {off1, off2, off3};
#pragma unroll
    for (int l = 0; l < 3; l++) {
        if ((uint32_t)l >= levels) break;
        const EF::Pre x = pt.x[n - 1 - (k + l)].pre();  // (uniform index: scalar loads from the argument segment)
        const int cnt = 1 << l;
#pragma unroll
        for (int u = cnt - 1; u >= 0; u--) {
            const EF lo = cur[u], hi = lo * x;
            cur[2 * u] = lo - hi;
            cur[2 * u + 1] = hi;
        }
        const uint64_t base = offs[l] + ((i << (l + 1)) - (dst_lo << (l + 1)));
#pragma unroll
        for (int u = 0; u < cnt; u++) stp2(heap.p, heap.stride, base + 2 * u, cur[2 * u], cur[2 * u + 1]);
    }
}
int eq_heap_run(const EF *point, const EqHeapPlan &pl, EFPlanes heap, hipStream_t s) {
    if (pl.n == 0 || pl.n > EqPoint::MAX) return (int)hipErrorInvalidValue;
    if (pl.empty) return 0;
    EqPoint d_point;
    for (uint32_t j = 0; j < pl.n; j++) d_point.x[j] = point[j];
    OVM_LAUNCH(eq_heap_top_kernel, dim3(ceil_div(2ull << pl.top, 256)), dim3(256), 0, s, d_point, pl.n, pl.top, heap);
    OVM_CHECK_LAUNCH();
    for (auto &ln : pl.launches) {
        const uint32_t k = ln.k;
        OVM_LAUNCH(eq_heap_levels_win_kernel, dim3(ceil_div(ln.count, 256)), dim3(256), 0, s, d_point, pl.n, k, ln.levels,
                           heap, pl.off[k], pl.lo[k], ln.src_first, ln.count, pl.off[k + 1], ln.levels > 1 ? pl.off[k + 2] : 0,
                           ln.levels > 2 ? pl.off[k + 3] : 0, ln.src_first);
        OVM_CHECK_LAUNCH();
    }
    return 0;
}

// out[2i], out[2i+1] = (p, q) of the fractions at plane positions pos[i] (n <= 32); pbase[i] != 0: only
// limb plane 0 of that numerator exists. No flag unless ro.flag.
__global__ void gkr_read_list_kernel(FracPlanes t, const uint64_t *pos, const uint32_t *pbase, uint32_t n, RedOut ro) {
    const uint32_t i = threadIdx.x;
    if (i < n) {
        const uint64_t at = pos[i];
        EF p = pbase[i] ? EF::from_base(Fp::raw(t.P()[at])) : ldp(t.P(), t.stride, at);
        st_ef_sys(ro.out + 2 * i, p);
        st_ef_sys(ro.out + 2 * i + 1, ldp(t.Q(), t.stride, at));
        release_to_host();
    }
    __syncthreads();
    if (i == 0) publish(ro);
}
int gkr_read_list(FracPlanes t, const uint64_t *d_pos, const uint32_t *d_pbase, uint32_t n, const RedOut &ro, hipStream_t s) {
    if (n > 64) return (int)hipErrorInvalidValue;
    OVM_LAUNCH(gkr_read_list_kernel, dim3(1), dim3(64), 0, s, t, d_pos, d_pbase, n, ro);
    OVM_CHECK_LAUNCH();
    return 0;
}

// Table records for the host: out[4k..4k+4) = (p0, q0, p1, q1) of record k. Records 0..n0-1 are consecutive
// starting at plane offset at0, records n0..n0+n1-1 consecutive from at1 (two unpaired edge records, or one
// run = a rank's whole remaining table). fold == 0: a record is the two fractions there; fold != 0: it is the
// fold by r of two records = four fractions (the gkr_fold_final of round 1, at any index).
// One fold step on an array of fractions: records (f[4i], f[4i+1]) and (f[4i+2], f[4i+3]) -> record (g[2i], g[2i+1]).
template <int N>
__device__ __forceinline__ void fold_fracs(const EF (&f)[N], EF (&g)[N / 2], const EF::Pre &r) {
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
        g[2 * i] = f[4 * i] + (f[4 * i + 2] - f[4 * i]) * r;
        g[2 * i + 1] = f[4 * i + 1] + (f[4 * i + 3] - f[4 * i + 1]) * r;
    }
}
// folds = 0: a record is the two fractions there; 1: the fold by r1 of two records (four fractions); 2: the fold by r2 of
// two such folds (eight fractions).
__device__ __forceinline__ void gkr_edge_body(uint32_t k, const FracPlanes &src, uint64_t at0, uint64_t at1, uint32_t n0, uint32_t n1,
                                              int p_base, int folds, const EF &r1_in, const EF &r2_in, EF *out_all) {
    if (k < n0 + n1) {
        const uint64_t step = 2ull << folds;
        const uint64_t at = k < n0 ? at0 + step * k : at1 + step * (k - n0);
        EF *out = out_all + 4 * k;
        if (folds == 2) {
            const EF::Pre r1 = r1_in.pre(), r2 = r2_in.pre();
            EF f[8], g[4], pq[2][2];
#pragma unroll
            for (int isq = 0; isq < 2; isq++) {
                EF lo[4], hi[4];
                if (!isq && p_base) {
                    ldp4_base(src.P(), at, lo);
                    ldp4_base(src.P(), at + 4, hi);
                } else {
                    ldp4(isq ? src.Q() : src.P(), src.stride, at, lo);
                    ldp4(isq ? src.Q() : src.P(), src.stride, at + 4, hi);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) f[i] = lo[i], f[4 + i] = hi[i];
                fold_fracs<8>(f, g, r1);
                fold_fracs<4>(g, pq[isq], r2);
            }
            st_ef_sys(out + 0, pq[0][0]);
            st_ef_sys(out + 1, pq[1][0]);
            st_ef_sys(out + 2, pq[0][1]);
            st_ef_sys(out + 3, pq[1][1]);
        } else if (folds == 1) {
            const EF r = r1_in;
            EF p[4], q[4];
            if (p_base)
                ldp4_base(src.P(), at, p);
            else
                ldp4(src.P(), src.stride, at, p);
            ldp4(src.Q(), src.stride, at, q);
            st_ef_sys(out + 0, p[0] + (p[2] - p[0]) * r);
            st_ef_sys(out + 1, q[0] + (q[2] - q[0]) * r);
            st_ef_sys(out + 2, p[1] + (p[3] - p[1]) * r);
            st_ef_sys(out + 3, q[1] + (q[3] - q[1]) * r);
        } else {
            EF p0, p1, q0, q1;
            if (p_base)
                ldp2_base(src.P(), at, p0, p1);
            else
                ldp2(src.P(), src.stride, at, p0, p1);
            ldp2(src.Q(), src.stride, at, q0, q1);
            st_ef_sys(out + 0, p0);
            st_ef_sys(out + 1, q0);
            st_ef_sys(out + 2, p1);
            st_ef_sys(out + 3, q1);
        }
        release_to_host();
    }
}
__global__ void gkr_edge_kernel(FracPlanes src, uint64_t at0, uint64_t at1, uint32_t n0, uint32_t n1, int p_base, int folds, EF r1_in,
                                EF r2_in, RedOut ro) {
    const uint32_t k = threadIdx.x;
    gkr_edge_body(k, src, at0, at1, n0, n1, p_base, folds, r1_in, r2_in, ro.out);
    __syncthreads();
    if (k == 0) publish(ro);
}
// The same records written by a sub-round's sums kernel (its workgroup 0, before its own pairs): in a many-AIR proof most
// sub-rounds have edge records, and a launch of their own was 3 us of kernel and 3.7 us of dispatch in front of a 9 us
// sums kernel, 258 times per 46-AIR proof. n0 + n1 = 0: none. The records are released before the workgroup's arrival at
// the reduction counter, the flag is raised after the last arrival.
int gkr_edge_records(FracPlanes src, uint64_t at0, uint64_t at1, uint32_t n0, uint32_t n1, bool p_base, int folds, EF r1, EF r2,
                     const RedOut &ro, hipStream_t s) {
    if (n0 + n1 == 0 || n0 + n1 > 64 || folds < 0 || folds > 2) return (int)hipErrorInvalidValue;
    OVM_LAUNCH(gkr_edge_kernel, dim3(1), dim3(64), 0, s, src, at0, at1, n0, n1, p_base ? 1 : 0, folds, r1, r2, ro);
    OVM_CHECK_LAUNCH();
    return 0;
}

// The weight of pair y (see EqSplit): the full table of a big layer is a gigabyte to build and as much again to read
// -- a quarter of the layer's traffic -- while two tables of the square root of its size stay in cache, at one extra
// extension-field product per pair.
__device__ __forceinline__ EF eq_weight(const EqSplit &eq, uint64_t y) {
    if (eq.lo_bits == EQ_WHOLE) return ldp(eq.lo.p, eq.lo.stride, y);
    const uint64_t gy = eq.y_first + y;
    if (eq.lo_bits == 0) return ldp(eq.hi.p, eq.hi.stride, gy);
    return ldp(eq.lo.p, eq.lo.stride, gy & ((1ull << eq.lo_bits) - 1)) * ldp(eq.hi.p, eq.hi.stride, gy >> eq.lo_bits);
}
template <bool PBASE, int NP>
__global__ __launch_bounds__(RED_THREADS) void gkr_round_kernel(FracPlanes rec, EqSplit eq, uint64_t n_pairs,
                                                                EF lambda_in, RedOut ro, GkrEdges ed) {
    if (ed.n0 + ed.n1 && blockIdx.x == 0 && threadIdx.x < 64)
        gkr_edge_body(threadIdx.x, ed.src, ed.at0, ed.at1, ed.n0, ed.n1, ed.p_base, ed.folds, ed.r1, ed.r2, ed.out);
    const uint64_t stride = (uint64_t)gridDim.x * RED_THREADS;
    const EF::Pre lambda = lambda_in.pre();
    EF sums[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) sums[i] = EF::zero();
    for (uint64_t y = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; y < n_pairs; y += stride) {
        EF p[4], q[4];  // fractions 4y..4y+3 = records 2y (entries 0,1) and 2y+1 (entries 2,3)
        if (PBASE)
            ldp4_base(rec.P(), 4 * y, p);
        else
            ldp4(rec.P(), rec.stride, 4 * y, p);
        ldp4(rec.Q(), rec.stride, 4 * y, q);
        const EF e = eq_weight(eq, y);
        EF a[4] = {p[0], q[0], p[1], q[1]};
        const EF b[4] = {p[2], q[2], p[3], q[3]};
        gkr_pair_sums<NP>(a, b, e, lambda, sums);
    }
    block_sum_out<NP>(sums, ro, (size_t)blockIdx.x * NP);
    finish_reduction<NP>(ro, gridDim.x, NP);
}
int gkr_round(FracPlanes rec, bool p_base, EqSplit eq, uint64_t n_pairs, EF lambda, uint32_t n_points,
              const RedOut &ro, hipStream_t s, const GkrEdges &ed) {
    if (n_points != 2 && n_points != 3) return (int)hipErrorInvalidValue;
    if (ed.n0 + ed.n1 > 64 || ed.folds < 0 || ed.folds > 2 || (ed.n0 + ed.n1 && !ed.out)) return (int)hipErrorInvalidValue;
    uint32_t grid = red_grid(n_pairs);
#define CALL(PB, NP)                                                                                             \
    OVM_LAUNCH((gkr_round_kernel<PB, NP>), dim3(grid), dim3(RED_THREADS), 0, s, rec, eq, n_pairs, lambda, \
                       red_inline(ro, grid, n_points), ed)
    if (p_base) {
        if (n_points == 2) CALL(true, 2); else CALL(true, 3);
    } else {
        if (n_points == 2) CALL(false, 2); else CALL(false, 3);
    }
#undef CALL
    OVM_CHECK_LAUNCH();
    return red_finish(ro, grid, grid, n_points, s);
}

// Fused "fold with r, then next round's sums": output pair y is built from input records
// 4y..4y+3 = fractions 8y..8y+7; the two folded records (fractions 4y..4y+3 of the output) are
// stored and immediately used for the degree-3 round sums with weight eq[y], so every sub-round
// after the first is ONE pass over the data.
template <bool PBASE, int NP>
__global__ __launch_bounds__(RED_THREADS) void gkr_fold_round_kernel(FracPlanes in, EqSplit eq, FracPlanes out,
                                                                     uint64_t n_pairs, EF r_in, EF lambda_in,
                                                                     RedOut ro, GkrEdges ed) {
    if (ed.n0 + ed.n1 && blockIdx.x == 0 && threadIdx.x < 64)
        gkr_edge_body(threadIdx.x, ed.src, ed.at0, ed.at1, ed.n0, ed.n1, ed.p_base, ed.folds, ed.r1, ed.r2, ed.out);
    const uint64_t stride = (uint64_t)gridDim.x * RED_THREADS;
    const EF::Pre r = r_in.pre(), lambda = lambda_in.pre();
    EF sums[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) sums[i] = EF::zero();
    for (uint64_t y = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; y < n_pairs; y += stride) {
        // all loads of the pair first (P, Q and the weight are three independent latencies otherwise:
        // the stores to `out` in between keep the compiler from hoisting them)
        EF fp[4], fq[4];
        EF up[4], vp[4], uq[4], vq[4];
        if (PBASE) {
            ldp4_base(in.P(), 8 * y, up);
            ldp4_base(in.P(), 8 * y + 4, vp);
        } else {
            ldp4(in.P(), in.stride, 8 * y, up);      // records 4y, 4y+1
            ldp4(in.P(), in.stride, 8 * y + 4, vp);  // records 4y+2, 4y+3
        }
        ldp4(in.Q(), in.stride, 8 * y, uq);
        ldp4(in.Q(), in.stride, 8 * y + 4, vq);
        const EF e = eq_weight(eq, y);
        fp[0] = up[0] + (up[2] - up[0]) * r;
        fp[1] = up[1] + (up[3] - up[1]) * r;
        fp[2] = vp[0] + (vp[2] - vp[0]) * r;
        fp[3] = vp[1] + (vp[3] - vp[1]) * r;
        stp4(out.P(), out.stride, 4 * y, fp);
        fq[0] = uq[0] + (uq[2] - uq[0]) * r;
        fq[1] = uq[1] + (uq[3] - uq[1]) * r;
        fq[2] = vq[0] + (vq[2] - vq[0]) * r;
        fq[3] = vq[1] + (vq[3] - vq[1]) * r;
        stp4(out.Q(), out.stride, 4 * y, fq);
        EF a[4] = {fp[0], fq[0], fp[1], fq[1]};
        const EF b[4] = {fp[2], fq[2], fp[3], fq[3]};
        gkr_pair_sums<NP>(a, b, e, lambda, sums);
    }
    block_sum_out<NP>(sums, ro, (size_t)blockIdx.x * NP);
    finish_reduction<NP>(ro, gridDim.x, NP);
}
int gkr_fold_round(FracPlanes in, bool in_p_base, EqSplit eq, FracPlanes out, uint64_t n_pairs, EF r, EF lambda,
                   uint32_t n_points, const RedOut &ro, hipStream_t s, const GkrEdges &ed) {
    if (n_points != 2 && n_points != 3) return (int)hipErrorInvalidValue;
    if (ed.n0 + ed.n1 > 64 || ed.folds < 0 || ed.folds > 2 || (ed.n0 + ed.n1 && !ed.out)) return (int)hipErrorInvalidValue;
    uint32_t grid = red_grid(n_pairs);
#define CALL(PB, NP)                                                                                               \
    OVM_LAUNCH((gkr_fold_round_kernel<PB, NP>), dim3(grid), dim3(RED_THREADS), 0, s, in, eq, out, n_pairs, r, \
                       lambda, red_inline(ro, grid, n_points), ed)
    if (in_p_base) {
        if (n_points == 2) CALL(true, 2); else CALL(true, 3);
    } else {
        if (n_points == 2) CALL(false, 2); else CALL(false, 3);
    }
#undef CALL
    OVM_CHECK_LAUNCH();
    return red_finish(ro, grid, grid, n_points, s);
}

// ---- stacked reduction -----------------------------------------------------------------------
__global__ __launch_bounds__(RED_THREADS) void stacked_round0_kernel(const Fp *base, uint32_t n_slices,
                                                                     uint32_t log_len, uint32_t l_skip,
                                                                     const EF *lam_eq, const EF *lam_rot,
                                                                     const EF *eq_cube, EF *partials) {
    __shared__ EF sh[RED_THREADS];
    const uint32_t N = 1u << l_skip;
    const uint64_t total = 1ull << log_len, stride = (uint64_t)gridDim.x * RED_THREADS;
    const uint64_t nx = total >> l_skip;
    EF sums[3] = {EF::zero(), EF::zero(), EF::zero()};
    // blockIdx.y takes a chunk of the slices (the sums are linear in them): a short window with thousands of
    // slices is otherwise a handful of lanes walking all of them one after the other
    const uint32_t per = (n_slices + gridDim.y - 1) / gridDim.y;
    const uint32_t i0 = blockIdx.y * per, i1 = i0 + per < n_slices ? i0 + per : n_slices;
    for (uint64_t p = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; p < total; p += stride) {
        Acc96 le, lr;
        le.clear();
        lr.clear();
        // Four slices' loads in flight per lane (unrolled by hand: the compiler leaves a loop around the multiply-add asm
        // alone, `#pragma unroll` or not, and then every load's latency is exposed -- the sums are exact integers, any order)
        const Fp *col = base + p;
        uint32_t i = i0;
        if (lam_rot) {
            for (; i + 4 <= i1; i += 4) {
                const uint32_t v0 = col[(uint64_t)i << log_len].v, v1 = col[(uint64_t)(i + 1) << log_len].v;
                const uint32_t v2 = col[(uint64_t)(i + 2) << log_len].v, v3 = col[(uint64_t)(i + 3) << log_len].v;
                mac_weight(le, lam_eq, i, v0), mac_weight(lr, lam_rot, i, v0);
                mac_weight(le, lam_eq, i + 1, v1), mac_weight(lr, lam_rot, i + 1, v1);
                mac_weight(le, lam_eq, i + 2, v2), mac_weight(lr, lam_rot, i + 2, v2);
                mac_weight(le, lam_eq, i + 3, v3), mac_weight(lr, lam_rot, i + 3, v3);
            }
            for (; i < i1; i++) {
                const uint32_t v = col[(uint64_t)i << log_len].v;
                mac_weight(le, lam_eq, i, v);
                mac_weight(lr, lam_rot, i, v);
            }
        } else {  // no slice of this window has a rotated opening: its weights are all zero
            for (; i + 4 <= i1; i += 4) {
                const uint32_t v0 = col[(uint64_t)i << log_len].v, v1 = col[(uint64_t)(i + 1) << log_len].v;
                const uint32_t v2 = col[(uint64_t)(i + 2) << log_len].v, v3 = col[(uint64_t)(i + 3) << log_len].v;
                mac_weight(le, lam_eq, i, v0);
                mac_weight(le, lam_eq, i + 1, v1);
                mac_weight(le, lam_eq, i + 2, v2);
                mac_weight(le, lam_eq, i + 3, v3);
            }
            for (; i < i1; i++) mac_weight(le, lam_eq, i, col[(uint64_t)i << log_len].v);
        }
        const EF qe = le.finish(), qr = lr.finish();
        uint64_t x = p >> l_skip, xp = x == 0 ? nx - 1 : x - 1;
        EF ec = ld_ef(eq_cube + x), ecp = ld_ef(eq_cube + xp);
        sums[0] = sums[0] + ec * qe;
        sums[1] = sums[1] + ec * qr;
        sums[2] = sums[2] + (ecp - ec) * qr;
    }
    for (uint32_t v = 0; v < 3; v++) {
        sh[threadIdx.x] = sums[v];
        __syncthreads();
        if (threadIdx.x < N) {
            EF acc = sh[threadIdx.x];
            for (uint32_t j = threadIdx.x + N; j < RED_THREADS; j += N) acc = acc + sh[j];
            // (N > RED_THREADS: the grid is a power of two >= N / RED_THREADS, so a lane's positions share one residue
            // mod N; the workgroup covers the residues from zbase on and zero-fills the rest of its record)
            const uint32_t zbase = N > RED_THREADS ? (blockIdx.x * RED_THREADS) & (N - 1) : 0;
            EF *rec = partials + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3 + v) * N;
            st_ef(rec + zbase + threadIdx.x, acc);
            for (uint32_t seg = 0; N > RED_THREADS && seg < N; seg += RED_THREADS)
                if (seg != zbase) st_ef(rec + seg + threadIdx.x, EF::zero());
        }
        __syncthreads();
    }
}
// slice chunks of stacked_round0 / stacked_combine for a window of 2^log_len rows: enough workgroups to
// occupy the machine (records: grid * chunks <= RED_BLOCKS), at least 16 slices each
uint32_t stacked_round0_chunks(uint32_t n_slices, uint32_t log_len) {
    const uint32_t grid = red_grid(1ull << log_len);
    uint32_t chunks = RED_BLOCKS / 2 / grid;
    if (chunks > n_slices / 16) chunks = n_slices / 16;
    return chunks < 1 ? 1 : chunks;
}
int stacked_round0(const Fp *base, uint32_t n_slices, uint32_t log_len, uint32_t l_skip, const EF *d_lam_eq,
                   const EF *d_lam_rot, const EF *d_eq_cube, const RedOut &ro, hipStream_t s) {
    if ((1u << l_skip) > 4 * RED_THREADS || log_len < l_skip) return (int)hipErrorInvalidValue;
    uint32_t grid = red_grid(1ull << log_len);  // a power of two >= 2^l_skip / RED_THREADS
    const uint32_t chunks = stacked_round0_chunks(n_slices, log_len);
    OVM_LAUNCH(stacked_round0_kernel, dim3(grid, chunks), dim3(RED_THREADS), 0, s, base, n_slices, log_len, l_skip,
                       d_lam_eq, d_lam_rot, d_eq_cube, ro.partials);
    OVM_CHECK_LAUNCH();
    return sum_partials(ro, grid * chunks, 3u << l_skip, s);
}

// out_eq[x] = sum_i coef_eq[i] * q_i[x], out_rot likewise: extension-field products with wave-uniform
// left factors, accumulated lazily -- limb k of c * v is sum_j v_j * m[k][j] with m[k][j] = c[k-j] for
// j <= k and 11 * c[k-j+4] otherwise; the seven distinct factors (c0..c3, 11 c1..11 c3) come from the
// host, and the 16 products of a multiplication go straight into the 96-bit accumulators.
struct CoefPre {
    uint32_t c[4];
    uint32_t w[4];  // w[k] = 11 * c[k] (w[0] unused)
};
__device__ __forceinline__ void mac_ef_ef(Acc96 &acc, const CoefPre *coefs, uint32_t idx, const EF &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 c = ((const OVM_CONST_AS uint4 *)coefs)[2 * idx], w = ((const OVM_CONST_AS uint4 *)coefs)[2 * idx + 1];
    const uint32_t cc[4] = {c.x, c.y, c.z, c.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) mac96(acc.lo[k], acc.hi[k], v.c[j].v, j <= k ? cc[k - j] : ww[k - j + 4]);
#endif
}
__global__ void stacked_combine_kernel(const EF *q, const uint64_t *slice_off, uint32_t n_slices, uint32_t log_len,
                                       const CoefPre *coef_eq, const CoefPre *coef_rot, EF *out_eq, EF *out_rot,
                                       EF *scratch) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (1ull << log_len)) return;
    // blockIdx.y takes a chunk of the slices; with more than one chunk the partial sums go to
    // scratch[(chunk * 2 + {eq, rot}) << log_len | x] and stacked_combine_reduce_kernel adds them up
    const uint32_t per = (n_slices + gridDim.y - 1) / gridDim.y;
    const uint32_t i0 = blockIdx.y * per, i1 = i0 + per < n_slices ? i0 + per : n_slices;
    Acc96 ae, ar;
    ae.clear();
    ar.clear();
    for (uint32_t i = i0; i < i1; i++) {
        const EF v = ld_ef(q + slice_off[i] + x);
        mac_ef_ef(ae, coef_eq, i, v);
        mac_ef_ef(ar, coef_rot, i, v);
    }
    if (gridDim.y == 1) {
        st_ef(out_eq + x, ae.finish());
        st_ef(out_rot + x, ar.finish());
    } else {
        st_ef(scratch + (((uint64_t)blockIdx.y * 2) << log_len) + x, ae.finish());
        st_ef(scratch + (((uint64_t)blockIdx.y * 2 + 1) << log_len) + x, ar.finish());
    }
}
__global__ void stacked_combine_reduce_kernel(const EF *scratch, uint32_t chunks, uint32_t log_len, EF *out_eq,
                                              EF *out_rot) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (1ull << log_len)) return;
    EF ae = EF::zero(), ar = EF::zero();
    for (uint32_t c = 0; c < chunks; c++) {
        ae = ae + ld_ef(scratch + (((uint64_t)c * 2) << log_len) + x);
        ar = ar + ld_ef(scratch + (((uint64_t)c * 2 + 1) << log_len) + x);
    }
    st_ef(out_eq + x, ae);
    st_ef(out_rot + x, ar);
}
int stacked_combine(const EF *q, const uint64_t *d_slice_off, uint32_t n_slices, uint32_t log_len,
                    const uint32_t *d_coef_eq_pre, const uint32_t *d_coef_rot_pre, EF *out_eq, EF *out_rot, EF *d_scratch,
                    uint32_t chunks, hipStream_t s) {
    if (4ull * n_slices >= bb::P || chunks < 1 || (chunks > 1 && !d_scratch)) return (int)hipErrorInvalidValue;
    const uint32_t gx = (uint32_t)ceil_div(1ull << log_len, 256);
    OVM_LAUNCH(stacked_combine_kernel, dim3(gx, chunks), dim3(256), 0, s, q, d_slice_off, n_slices, log_len,
                       (const CoefPre *)d_coef_eq_pre, (const CoefPre *)d_coef_rot_pre, out_eq, out_rot, d_scratch);
    if (chunks > 1)
        OVM_LAUNCH(stacked_combine_reduce_kernel, dim3(gx), dim3(256), 0, s, d_scratch, chunks, log_len, out_eq,
                           out_rot);
    OVM_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(RED_THREADS) void stacked_round_kernel(const EF *vec4, uint32_t log_h, RedOut ro) {
    const uint64_t h = 1ull << log_h, n_y = h >> 1, stride = (uint64_t)gridDim.x * RED_THREADS;
    const EF *qe = vec4, *qr = vec4 + h, *eq = vec4 + 2 * h, *kr = vec4 + 3 * h;
    EF sums[2] = {EF::zero(), EF::zero()};
    for (uint64_t y = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; y < n_y; y += stride) {
        EF a0 = ld_ef(qe + 2 * y), a1 = ld_ef(qe + 2 * y + 1), b0 = ld_ef(qr + 2 * y), b1 = ld_ef(qr + 2 * y + 1);
        EF e0 = ld_ef(eq + 2 * y), e1 = ld_ef(eq + 2 * y + 1), k0 = ld_ef(kr + 2 * y), k1 = ld_ef(kr + 2 * y + 1);
        // X = 1
        sums[0] = sums[0] + a1 * e1 + b1 * k1;
        // X = 2: v(2) = 2 v1 - v0
        EF a2 = a1.dbl() - a0, b2 = b1.dbl() - b0, e2 = e1.dbl() - e0, k2 = k1.dbl() - k0;
        sums[1] = sums[1] + a2 * e2 + b2 * k2;
    }
    block_sum_out<2>(sums, ro, (size_t)blockIdx.x * 2);
    finish_reduction<2>(ro, gridDim.x, 2);
}
int stacked_round(const EF *vec4, uint32_t log_h, const RedOut &ro, hipStream_t s) {
    uint32_t grid = red_grid(1ull << (log_h - 1));
    OVM_LAUNCH(stacked_round_kernel, dim3(grid), dim3(RED_THREADS), 0, s, vec4, log_h, red_inline(ro, grid, 2));
    OVM_CHECK_LAUNCH();
    return red_finish(ro, grid, grid, 2, s);
}

// The finished slices of a stacked-reduction round (prover.hip: prove_stacked_reduction). A slice whose own variables are
// all bound contributes  q(X) * eq1b(X, b) * D_i  to the round polynomial, with q's two entries (t0, t1) read from the
// folded stacked matrix and D_i constant in X:  s(1) += b ? t1 D_i : 0,  s(2) += b ? 2 v : -v,  v = (2 t1 - t0) D_i;
// after the round D_i picks up eq1b(u_round, b). A many-AIR proof has ~13 000 such slices from the first rounds on, and
// the host looped over all of them every round with the device idle (gather + download + ~175 us of host products per
// round, 20 rounds). Here d[i] lives on the device: this kernel first applies the PREVIOUS round's factor (u_prev) to
// the slices that were already finished then, then sums the round's two values.
__global__ __launch_bounds__(RED_THREADS) void stacked_tail_kernel(const EF *q, uint32_t q_log_h, const TailSlice *slices, EF *d,
                                                                   uint32_t n, uint32_t l_skip, uint32_t round, EF u_prev_in,
                                                                   RedOut ro) {
    const EF::Pre u1 = u_prev_in.pre(), u0 = (EF::one() - u_prev_in).pre();
    EF sums[2] = {EF::zero(), EF::zero()};
    const uint32_t stride = gridDim.x * RED_THREADS;
    for (uint32_t i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += stride) {
        const TailSlice sl = slices[i];
        if (round <= sl.n_lift) continue;  // still has variables of its own (or is another rank's: n_lift = ~0)
        EF di = ld_ef(d + i);
        if (round - 1 > sl.n_lift) {
            di = ((sl.row >> (l_skip + round - 2)) & 1) ? di * u1 : di * u0;
            st_ef(d + i, di);
        }
        const uint64_t at = ((uint64_t)sl.gcol << q_log_h) + ((sl.row >> (l_skip + round)) << 1);
        const EF t0 = ld_ef(q + at), t1 = ld_ef(q + at + 1);
        const EF::Pre dp = di.pre();
        const EF v = (t1 + t1 - t0) * dp;
        if ((sl.row >> (l_skip + round - 1)) & 1) {
            sums[0] = sums[0] + t1 * dp;
            sums[1] = sums[1] + v + v;
        } else {
            sums[1] = sums[1] - v;
        }
    }
    block_sum_out<2>(sums, ro, (size_t)blockIdx.x * 2);
    finish_reduction<2>(ro, gridDim.x, 2);
}
int stacked_tail(const EF *q, uint32_t q_log_h, const TailSlice *d_slices, EF *d_tail, uint32_t n, uint32_t l_skip, uint32_t round,
                 EF u_prev, const RedOut &ro, hipStream_t s) {
    if (n == 0 || round == 0) return (int)hipErrorInvalidValue;
    uint32_t grid = red_grid(n);
    if (grid > tuning().finish_inline_max) grid = tuning().finish_inline_max;  // (always finished in the kernel)
    OVM_LAUNCH(stacked_tail_kernel, dim3(grid), dim3(RED_THREADS), 0, s, q, q_log_h, d_slices, d_tail, n, l_skip, round, u_prev,
               red_inline(ro, grid, 2));
    OVM_CHECK_LAUNCH();
    return 0;
}

// ---- WHIR ------------------------------------------------------------------------------------
__global__ void whir_mu_batch_kernel(const Fp *msg, uint32_t width, uint32_t log_h, const EF *mu_pows,
                                     uint32_t mu_offset, EF *f, int accumulate) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1ull << log_h)) return;
    Acc96 lazy;
    lazy.clear();
    // (four columns' loads in flight per lane: unrolled by hand, see stacked_round0_kernel)
    const Fp *col = msg + i;
    uint32_t j = 0;
    for (; j + 4 <= width; j += 4) {
        const uint32_t v0 = col[(uint64_t)j << log_h].v, v1 = col[(uint64_t)(j + 1) << log_h].v;
        const uint32_t v2 = col[(uint64_t)(j + 2) << log_h].v, v3 = col[(uint64_t)(j + 3) << log_h].v;
        mac_weight(lazy, mu_pows, mu_offset + j, v0);
        mac_weight(lazy, mu_pows, mu_offset + j + 1, v1);
        mac_weight(lazy, mu_pows, mu_offset + j + 2, v2);
        mac_weight(lazy, mu_pows, mu_offset + j + 3, v3);
    }
    for (; j < width; j++) mac_weight(lazy, mu_pows, mu_offset + j, col[(uint64_t)j << log_h].v);
    const EF sum = lazy.finish();
    st_ef(f + i, accumulate ? ld_ef(f + i) + sum : sum);
}
int whir_mu_batch(const Fp *msg, uint32_t width, uint32_t log_h, const EF *d_mu_pows, uint32_t mu_offset, EF *f,
                  int accumulate, hipStream_t s) {
    OVM_LAUNCH(whir_mu_batch_kernel, dim3(ceil_div(1ull << log_h, 256)), dim3(256), 0, s, msg, width, log_h,
                       d_mu_pows, mu_offset, f, accumulate);
    OVM_CHECK_LAUNCH();
    return 0;
}

// One pass of up to ZT_TILE_LOG butterfly stages (bits [b0, b0+nb)) of the subset-zeta / Moebius
// transform, staged through LDS: tile = 2^nb x 2^c elements, c low bits contiguous. The stages run FOUR AT A TIME in
// registers: a lane takes the 16 elements that differ in four tile bits, applies the four stages (they act on
// different bits, so their order does not matter: exact field additions) and puts them back -- one LDS round trip and
// one barrier per four stages instead of per stage (the per-stage form moved 24 bytes of LDS per element per stage and
// ran at 1.7 TB/s of HBM: 0.66 ms for the 2^24-point transform of the mu-batch). The tile is padded by one element
// per 16, like the NTT's (a lane reading 16 consecutive elements / lanes reading consecutive elements: both
// conflict-free).
static constexpr uint32_t ZT_TILE_LOG = 11;
__global__ __launch_bounds__(256) void zeta_pass_kernel(EF *v, uint32_t log_n, uint32_t b0, uint32_t nb, uint32_t c,
                                                        int dir) {
    extern __shared__ unsigned char zsm[];
    EF *tile = reinterpret_cast<EF *>(zsm);
    auto pad = [](uint32_t l) -> uint32_t { return l + (l >> 4); };
    const uint32_t tile_size = 1u << (nb + c), c_mask = (1u << c) - 1;
    // position = (hi << (b0 + nb)) | (t << b0) | (mid << c) | cc   (c <= b0)
    const uint64_t tile_id = blockIdx.x;
    const uint64_t mid = tile_id & ((1ull << (b0 - c)) - 1), hi = tile_id >> (b0 - c);
    const uint64_t base = (hi << (b0 + nb)) | (mid << c);
    for (uint32_t l = threadIdx.x; l < tile_size; l += 256) {
        uint64_t p = base | ((uint64_t)(l >> c) << b0) | (l & c_mask);
        tile[pad(l)] = ld_ef(v + p);
    }
    __syncthreads();
    uint32_t k = 0;
    for (; k + 4 <= nb; k += 4) {
        const uint32_t lbit = k + c, lo_mask = (1u << lbit) - 1;
        for (uint32_t g = threadIdx.x; g < (tile_size >> 4); g += 256) {
            const uint32_t l_base = ((g & ~lo_mask) << 4) | (g & lo_mask);
            EF x[16];
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = tile[pad(l_base | ((uint32_t)e << lbit))];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (e >> j & 1) x[e] = dir > 0 ? x[e] + x[e ^ (1 << j)] : x[e] - x[e ^ (1 << j)];
#pragma unroll
            for (int e = 0; e < 16; e++) tile[pad(l_base | ((uint32_t)e << lbit))] = x[e];
        }
        __syncthreads();
    }
    for (; k < nb; k++) {
        const uint32_t lbit = k + c, lo_mask = (1u << lbit) - 1;
        for (uint32_t bf = threadIdx.x; bf < (tile_size >> 1); bf += 256) {
            uint32_t l0 = ((bf & ~lo_mask) << 1) | (bf & lo_mask), l1 = l0 | (1u << lbit);
            tile[pad(l1)] = dir > 0 ? tile[pad(l1)] + tile[pad(l0)] : tile[pad(l1)] - tile[pad(l0)];
        }
        __syncthreads();
    }
    for (uint32_t l = threadIdx.x; l < tile_size; l += 256) {
        uint64_t p = base | ((uint64_t)(l >> c) << b0) | (l & c_mask);
        st_ef(v + p, tile[pad(l)]);
    }
}
int zeta_transform(EF *v, uint32_t log_n, int dir, hipStream_t s) {
    // low bits first in one contiguous pass, then strided passes of up to ZT_TILE_LOG-4 bits
    uint32_t b = 0;
    while (b < log_n) {
        uint32_t nb, c;
        if (b == 0) {
            nb = log_n < ZT_TILE_LOG ? log_n : ZT_TILE_LOG;
            c = 0;
        } else {
            c = 4;  // 16 EF = 256 B runs
            nb = log_n - b < ZT_TILE_LOG - c ? log_n - b : ZT_TILE_LOG - c;
        }
        uint32_t tiles = 1u << (log_n - nb - c);
        OVM_LAUNCH(zeta_pass_kernel, dim3(tiles), dim3(256), sizeof(EF) * ((1u << (nb + c)) + (1u << (nb + c) >> 4) + 1), s, v, log_n, b, nb,
                   c, dir);
        OVM_CHECK_LAUNCH();
        b += nb;
    }
    return 0;
}

__global__ __launch_bounds__(RED_THREADS) void whir_round_kernel(const EF *f, const EF *w, uint64_t n_pairs,
                                                                 RedOut ro) {
    const uint64_t stride = (uint64_t)gridDim.x * RED_THREADS;
    EF sums[2] = {EF::zero(), EF::zero()};
    for (uint64_t y = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; y < n_pairs; y += stride) {
        EF f0 = ld_ef(f + 2 * y), f1 = ld_ef(f + 2 * y + 1), w0 = ld_ef(w + 2 * y), w1 = ld_ef(w + 2 * y + 1);
        sums[0] = sums[0] + f1 * w1;
        sums[1] = sums[1] + (f1.dbl() - f0) * (w1.dbl() - w0);
    }
    block_sum_out<2>(sums, ro, (size_t)blockIdx.x * 2);
    finish_reduction<2>(ro, gridDim.x, 2);
}
int whir_round(const EF *f, const EF *w, uint64_t n_pairs, const RedOut &ro, hipStream_t s) {
    uint32_t grid = red_grid(n_pairs);
    OVM_LAUNCH(whir_round_kernel, dim3(grid), dim3(RED_THREADS), 0, s, f, w, n_pairs, red_inline(ro, grid, 2));
    OVM_CHECK_LAUNCH();
    return red_finish(ro, grid, grid, 2, s);
}

// Fused "fold by alpha, then the next round's sums" (as in the GKR phase): output pair y is built
// from f[4y..4y+3], w[4y..4y+3]; one pass over the data per sumcheck round instead of two.
__global__ __launch_bounds__(RED_THREADS) void whir_fold_round_kernel(const EF *f_in, const EF *w_in, EF *f_out,
                                                                      EF *w_out, uint64_t n_pairs, EF alpha_in,
                                                                      RedOut ro) {
    const uint64_t stride = (uint64_t)gridDim.x * RED_THREADS;
    const EF::Pre alpha = alpha_in.pre();
    EF sums[2] = {EF::zero(), EF::zero()};
    for (uint64_t y = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; y < n_pairs; y += stride) {
        EF a0 = ld_ef(f_in + 4 * y), a1 = ld_ef(f_in + 4 * y + 1), a2 = ld_ef(f_in + 4 * y + 2), a3 = ld_ef(f_in + 4 * y + 3);
        EF b0 = ld_ef(w_in + 4 * y), b1 = ld_ef(w_in + 4 * y + 1), b2 = ld_ef(w_in + 4 * y + 2), b3 = ld_ef(w_in + 4 * y + 3);
        const EF f0 = a0 + (a1 - a0) * alpha, f1 = a2 + (a3 - a2) * alpha;
        const EF w0 = b0 + (b1 - b0) * alpha, w1 = b2 + (b3 - b2) * alpha;
        st_ef(f_out + 2 * y, f0);
        st_ef(f_out + 2 * y + 1, f1);
        st_ef(w_out + 2 * y, w0);
        st_ef(w_out + 2 * y + 1, w1);
        sums[0] = sums[0] + f1 * w1;
        sums[1] = sums[1] + (f1.dbl() - f0) * (w1.dbl() - w0);
    }
    block_sum_out<2>(sums, ro, (size_t)blockIdx.x * 2);
    finish_reduction<2>(ro, gridDim.x, 2);
}
int whir_fold_round(const EF *f_in, const EF *w_in, EF *f_out, EF *w_out, uint64_t n_pairs, EF alpha, const RedOut &ro,
                    hipStream_t s) {
    uint32_t grid = red_grid(n_pairs);
    OVM_LAUNCH(whir_fold_round_kernel, dim3(grid), dim3(RED_THREADS), 0, s, f_in, w_in, f_out, w_out, n_pairs, alpha,
                       red_inline(ro, grid, 2));
    OVM_CHECK_LAUNCH();
    return red_finish(ro, grid, grid, 2, s);
}

__global__ void whir_fold_kernel(const EF *f_in, const EF *w_in, EF *f_out, EF *w_out, uint64_t n_out, EF alpha_in) {
    uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= n_out) return;
    const EF::Pre alpha = alpha_in.pre();
    EF f0 = ld_ef(f_in + 2 * y), f1 = ld_ef(f_in + 2 * y + 1), w0 = ld_ef(w_in + 2 * y), w1 = ld_ef(w_in + 2 * y + 1);
    st_ef(f_out + y, f0 + (f1 - f0) * alpha);
    st_ef(w_out + y, w0 + (w1 - w0) * alpha);
}
int whir_fold(const EF *f_in, const EF *w_in, EF *f_out, EF *w_out, uint64_t n_out, EF alpha, hipStream_t s) {
    OVM_LAUNCH(whir_fold_kernel, dim3(ceil_div(n_out, 256)), dim3(256), 0, s, f_in, w_in, f_out, w_out, n_out,
                       alpha);
    OVM_CHECK_LAUNCH();
    return 0;
}

// sum_i c_i z^i: each lane Horner-evaluates a contiguous run of HORNER_RUN coefficients and scales
// by z^(run start).
static constexpr uint32_t HORNER_RUN = 16;
__global__ __launch_bounds__(RED_THREADS) void horner_kernel(const EF *coeffs, uint64_t n, EF z, EF z_run,
                                                             RedOut ro) {
    const uint64_t n_runs = (n + HORNER_RUN - 1) / HORNER_RUN, stride = (uint64_t)gridDim.x * RED_THREADS;
    EF sums[1] = {EF::zero()};
    for (uint64_t r = (uint64_t)blockIdx.x * RED_THREADS + threadIdx.x; r < n_runs; r += stride) {
        uint64_t lo = r * HORNER_RUN, hi = lo + HORNER_RUN < n ? lo + HORNER_RUN : n;
        EF acc = EF::zero();
        for (uint64_t i = hi; i-- > lo;) acc = acc * z + ld_ef(coeffs + i);
        sums[0] = sums[0] + acc * z_run.pow(r);
    }
    block_sum_out<1>(sums, ro, blockIdx.x);
    finish_reduction<1>(ro, gridDim.x, 1);
}
int horner_eval(const EF *coeffs, uint64_t n, EF z, const RedOut &ro, hipStream_t s) {
    uint32_t grid = red_grid((n + HORNER_RUN - 1) / HORNER_RUN);
    OVM_LAUNCH(horner_kernel, dim3(grid), dim3(RED_THREADS), 0, s, coeffs, n, z, z.pow(HORNER_RUN),
                       red_inline(ro, grid, 1));
    OVM_CHECK_LAUNCH();
    return red_finish(ro, grid, grid, 1, s);
}

// tab_lo[q * 2^log_lo + x_lo] (already scaled by gamma_q), tab_hi[q * 2^(log_n - log_lo) + x_hi]
__global__ void whir_w_accumulate_kernel(EF *w, uint32_t log_n, const EF *tab_lo, const EF *tab_hi, uint32_t log_lo,
                                         uint32_t n_q) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (1ull << log_n)) return;
    uint64_t lo = x & ((1ull << log_lo) - 1), hi = x >> log_lo;
    const uint32_t log_hi = log_n - log_lo;
    EF acc = ld_ef(w + x);
    for (uint32_t q = 0; q < n_q; q++)
        acc = acc + ld_ef(tab_lo + ((uint64_t)q << log_lo) + lo) * ld_ef(tab_hi + ((uint64_t)q << log_hi) + hi);
    st_ef(w + x, acc);
}
// The same sum with the products left unreduced: for log_lo >= 6 a wave's 64 positions share x_hi, so tab_hi's entry is
// wave-uniform (scalar loads) and  sum_q a_q * b_q  over F[X]/(X^4 - 11) is 16 multiply-adds per q into SEVEN 96-bit
// accumulators (the coefficients of X^0..X^6 of the plain polynomial product), reduced once after the loop and folded
// with X^4 = 11 -- instead of, per q, three products for 11 * b, four Montgomery reductions and an extension-field
// addition (~100 issue slots against 48; the n_q = 194 queries of a config D round: 458 -> ~230 us).
__global__ __launch_bounds__(256) void whir_w_accumulate_lazy_kernel(EF *w, uint32_t log_n, const EF *tab_lo, const EF *tab_hi,
                                                                     uint32_t log_lo, uint32_t n_q) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // (the grid is exactly 2^log_n / 256)
    const uint32_t lo = (uint32_t)(x & ((1ull << log_lo) - 1));
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> log_lo));
    const uint32_t log_hi = log_n - log_lo;
    uint64_t alo[7];
    uint32_t ahi[7];
#pragma unroll
    for (int k = 0; k < 7; k++) alo[k] = 0, ahi[k] = 0;
    for (uint32_t q = 0; q < n_q; q++) {
        const EF a = ld_ef(tab_lo + ((uint64_t)q << log_lo) + lo);
        const uint4 b4 = ((const OVM_CONST_AS uint4 *)tab_hi)[((uint64_t)q << log_hi) + hi];
        const uint32_t b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) mac96(alo[i + j], ahi[i + j], a.c[i].v, b[j]);
    }
    Fp f[7];
#pragma unroll
    for (int k = 0; k < 7; k++) {  // (hi * 2^64 + lo) * 2^-32 mod P, as Acc96::finish
        uint32_t xh = (uint32_t)(alo[k] >> 32);
        xh = bb::umin32(xh, xh - bb::P);
        xh = bb::umin32(xh, xh - bb::P);
        f[k] = Fp::raw(bb::mont_reduce(((uint64_t)xh << 32) | (uint32_t)alo[k])) + Fp::raw(ahi[k]) * Fp::raw(bb::R2);
    }
    const Fp W = bb::fp_const<11>();
    EF acc = ld_ef(w + x);
    acc.c[0] += f[0] + f[4] * W;
    acc.c[1] += f[1] + f[5] * W;
    acc.c[2] += f[2] + f[6] * W;
    acc.c[3] += f[3];
    st_ef(w + x, acc);
#endif
}
int whir_w_accumulate(EF *w, uint32_t log_n, const EF *d_tab_lo, const EF *d_tab_hi, uint32_t log_lo, uint32_t n_q,
                      hipStream_t s) {
    if (log_lo >= 6 && log_n >= 8 && tuning().lazy_w_accumulate) {
        OVM_LAUNCH(whir_w_accumulate_lazy_kernel, dim3(1u << (log_n - 8)), dim3(256), 0, s, w, log_n, d_tab_lo, d_tab_hi, log_lo, n_q);
        OVM_CHECK_LAUNCH();
        return 0;
    }
    OVM_LAUNCH(whir_w_accumulate_kernel, dim3(ceil_div(1ull << log_n, 256)), dim3(256), 0, s, w, log_n,
                       d_tab_lo, d_tab_hi, log_lo, n_q);
    OVM_CHECK_LAUNCH();
    return 0;
}

__global__ void gather_ef_kernel(const EF *src, const uint64_t *idx, uint64_t n, EF *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st_ef(out + i, ld_ef(src + idx[i]));
}
int gather_ef(const EF *src, const uint64_t *d_idx, uint64_t n, EF *out, hipStream_t s) {
    if (n == 0) return 0;
    OVM_LAUNCH(gather_ef_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, src, d_idx, n, out);
    OVM_CHECK_LAUNCH();
    return 0;
}

// out[i] = sum over the n_blocks vectors in[b * n + i] (the per-rank mu-batched WHIR vectors of a sharded proof,
// all-gathered; `out` may be one of the blocks' source buffers but not inside `in`)
__global__ void ef_sum_blocks_kernel(const EF *in, uint32_t n_blocks, uint64_t n, EF *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    EF acc = ld_ef(in + i);
    for (uint32_t b = 1; b < n_blocks; b++) acc = acc + ld_ef(in + (uint64_t)b * n + i);
    st_ef(out + i, acc);
}
int ef_sum_blocks(const EF *in, uint32_t n_blocks, uint64_t n, EF *out, hipStream_t s) {
    if (n_blocks == 0) return (int)hipErrorInvalidValue;
    OVM_LAUNCH(ef_sum_blocks_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, n_blocks, n, out);
    OVM_CHECK_LAUNCH();
    return 0;
}

}  // namespace ovm
