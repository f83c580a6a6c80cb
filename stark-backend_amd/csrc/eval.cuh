// Constraint-evaluation kernel bodies, parameterised by the program that evaluates one row:
// the register-program interpreter and the staged VM (stages.hip). Device code only.
#pragma once
#include "device_common.h"

namespace ovm {

// ---- extension-field memory access (16-byte vectors) ----------------------------------------
__device__ __forceinline__ EF ld_ef(const EF *p) {
    uint4 v = *reinterpret_cast<const uint4 *>(p);
    EF r;
    r.c[0] = Fp::raw(v.x);
    r.c[1] = Fp::raw(v.y);
    r.c[2] = Fp::raw(v.z);
    r.c[3] = Fp::raw(v.w);
    return r;
}
__device__ __forceinline__ void st_ef(EF *p, const EF &x) {
    *reinterpret_cast<uint4 *>(p) = make_uint4(x.c[0].v, x.c[1].v, x.c[2].v, x.c[3].v);
}
// (write-through forms for mailboxes and partial records: see release_to_host / release_to_device)
__device__ __forceinline__ void st_ef_sys(EF *p, const EF &x) {
#pragma unroll
    for (int k = 0; k < 4; k++) st_word_sys(&p->c[k].v, x.c[k].v);
}
__device__ __forceinline__ EF ld_ef_agent(const EF *p) {
    EF r;
#pragma unroll
    for (int k = 0; k < 4; k++) r.c[k] = Fp::raw(ld_word_agent(&p->c[k].v));
    return r;
}
__device__ __forceinline__ void st_ef_agent(EF *p, const EF &x) {
#pragma unroll
    for (int k = 0; k < 4; k++) st_word_agent(&p->c[k].v, x.c[k].v);
}
__device__ __forceinline__ void stp(uint32_t *b, uint64_t st, uint64_t i, const EF &x) {
#pragma unroll
    for (int k = 0; k < 4; k++) b[k * st + i] = x.c[k].v;
}

// ---- lazily reduced sums of weight * cell --------------------------------------------------------
// value = hi * 2^64 + lo per limb, a sum of raw 62-bit Montgomery products; reduced once by finish().
struct Acc96 {
    uint64_t lo[4];
    uint32_t hi[4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 4; k++) lo[k] = 0, hi[k] = 0;
    }
    __device__ __forceinline__ void add(const EF &w, Fp a) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t prod = (uint64_t)a.v * w.c[k].v;
            lo[k] += prod;
            hi[k] += lo[k] < prod;
        }
    }
    // (hi * 2^64 + lo) * 2^-32 mod P per limb
    __device__ __forceinline__ EF finish() const {
        EF r;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t xh = (uint32_t)(lo[k] >> 32);
            xh = bb::umin32(xh, xh - bb::P);  // < 2^32 < 3P: two steps bring it below P
            xh = bb::umin32(xh, xh - bb::P);
            const Fp low = Fp::raw(bb::mont_reduce(((uint64_t)xh << 32) | (uint32_t)lo[k]));
            r.c[k] = low + Fp::raw(hi[k]) * Fp::raw(bb::R2);  // hi < n_instr < P
        }
        return r;
    }
};

// acc += w * a for a wave-uniform extension-field weight (in SGPRs) and a base-field value: four
// 96-bit multiply-adds (v_mad_u64_u32 with carry-out + v_addc) instead of four Montgomery products
// and four modular additions; one reduction when the sum is finished (Acc96::finish). Used wherever
// a kernel sums weight_i * cell_i over many cells.
__device__ __forceinline__ void mac96(uint64_t &lo, uint32_t &hi, uint32_t a, uint32_t w) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "s"(w)
        : "vcc");
}
#define OVM_CONST_AS __attribute__((address_space(4)))
__device__ __forceinline__ void mac_weight(Acc96 &acc, const EF *weights, uint32_t idx, uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 w = ((const OVM_CONST_AS uint4 *)weights)[idx];  // scalar load: the weight table is read-only here
    mac96(acc.lo[0], acc.hi[0], a, w.x);
    mac96(acc.lo[1], acc.hi[1], a, w.y);
    mac96(acc.lo[2], acc.hi[2], a, w.z);
    mac96(acc.lo[3], acc.hi[3], a, w.w);
#endif
}

// ---- round 0: program values on the coset LDEs, weighted by eq and summed per (coset, z) ------
// A lane evaluates Prog::ROWS rows per step (rows p, p + stride, ...: same z = p mod N).
// Prog::operator()(const Fp *lde_k, uint32_t log_lifted, const uint64_t (&p)[ROWS], EF (&out)[ROWS][2])
// adds the program's accumulators at those rows of coset k (lde_k = column 0 of that coset).
//   partials[((block * n_cosets + k) * n_out + o) * N + z]
template <class Prog>
__device__ __forceinline__ void round0_eval_body(const Prog &prog, const Fp *lde, uint32_t n_cols_total,
                                                 uint32_t log_lifted, uint32_t l_skip, uint32_t n_cosets,
                                                 uint32_t n_out, const EF *eq_xi, EF add_const, EF *partials,
                                                 uint32_t bx = blockIdx.x, uint32_t gx = gridDim.x,
                                                 bool fence = false) {
    // (bx, gx: this workgroup's index and the number of workgroups of ITS job -- a batched launch runs
    // several jobs side by side; fence: device-scope release of the partial records for an in-kernel finish)
    constexpr int R = Prog::ROWS;
    // Scratch of the final reduction over the workgroup. A program that keeps an LDS image of its own (cached_vm.cuh)
    // lends it: 4 KiB of static LDS on top of a 32 KiB image is the difference between four and five workgroups per CU.
    EF *sh;
    if constexpr (Prog::REDUCE_IN_IMAGE) {
        extern __shared__ uint32_t lds_image[];
        sh = reinterpret_cast<EF *>(lds_image);
    } else {
        __shared__ EF sh_static[RED_THREADS];
        sh = sh_static;
    }
    const uint32_t N = 1u << l_skip;
    const uint64_t total = 1ull << log_lifted;
    // A lane's rows all have the same z = row mod N: the stride is a multiple of N -- for N <= RED_THREADS always, for a
    // larger skip domain (l_skip 9: N = 512) because the launcher makes gx a multiple of N / RED_THREADS. A workgroup then
    // covers the RED_THREADS residues from zbase on and writes zeros for the rest of its record.
    const uint64_t stride = (uint64_t)gx * RED_THREADS;
    const uint32_t zbase = N > RED_THREADS ? (uint32_t)(((uint64_t)bx * RED_THREADS) & (N - 1)) : 0;
    EF sums[MAX_COSETS * 2];
#pragma unroll
    for (int i = 0; i < (int)(MAX_COSETS * 2); i++) sums[i] = EF::zero();
    for (uint64_t p0 = (uint64_t)bx * RED_THREADS + threadIdx.x; p0 < total; p0 += stride * R) {
        uint64_t p[R];
        bool live[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint64_t q = p0 + (uint64_t)r * stride;
            live[r] = q < total;  // a row past the end is evaluated at p0 with weight zero
            p[r] = live[r] ? q : p0;
        }
        // (a runtime loop on purpose: unrolling it would replicate the whole row program per coset)
        for (uint32_t k = 0; k < n_cosets; k++) {
            EF out[R][2];
#pragma unroll
            for (int r = 0; r < R; r++) {
                out[r][0] = EF::zero();
                out[r][1] = add_const;
            }
            prog(lde + (((uint64_t)k * n_cols_total) << log_lifted), log_lifted, p, out);
#pragma unroll
            for (int r = 0; r < R; r++) {
                // (loaded here, not held across the program: 7 registers per row)
                const EF::Pre eq = (live[r] ? ld_ef(eq_xi + (p[r] >> l_skip)) : EF::zero()).pre();
                sums[2 * k] = sums[2 * k] + out[r][0] * eq;
                if (n_out > 1) sums[2 * k + 1] = sums[2 * k + 1] + out[r][1] * eq;
            }
        }
    }
    // reduce over lanes with equal z = tid mod N
    if constexpr (Prog::REDUCE_IN_IMAGE) __syncthreads();  // (every wave is done with the image before it is overwritten)
    for (uint32_t k = 0; k < n_cosets; k++)
        for (uint32_t o = 0; o < n_out; o++) {
            sh[threadIdx.x] = sums[2 * k + o];
            __syncthreads();
            if (threadIdx.x < N) {
                EF acc = sh[threadIdx.x];
                for (uint32_t j = threadIdx.x + N; j < RED_THREADS; j += N) acc = acc + sh[j];
                EF *rec = partials + ((size_t)bx * n_cosets * n_out + k * n_out + o) * N;
                if (fence) {
                    st_ef_agent(rec + zbase + threadIdx.x, acc);
                    for (uint32_t seg = 0; N > RED_THREADS && seg < N; seg += RED_THREADS)
                        if (seg != zbase) st_ef_agent(rec + seg + threadIdx.x, EF::zero());
                    release_to_device();
                } else {
                    st_ef(rec + zbase + threadIdx.x, acc);
                    for (uint32_t seg = 0; N > RED_THREADS && seg < N; seg += RED_THREADS)
                        if (seg != zbase) st_ef(rec + seg + threadIdx.x, EF::zero());
                }
            }
            __syncthreads();
        }
}

// ---- LogUp-GKR leaves ---------------------------------------------------------------------------
struct BaseFetch {
    const ColDesc *cols;
    uint64_t row;
    __device__ __forceinline__ Fp operator()(uint32_t col) const {
        const ColDesc cd = cols[col];
        return cd.ptr[(row + cd.rot) & ((1ull << cd.log_height) - 1)];
    }
};
struct LeafEmit {
    FracPlanes leaves;
    const uint64_t *leaf_base;
    const uint32_t *leaf_log_len;
    uint32_t log_height;
    uint64_t row;
    Fp norm;
    __device__ __forceinline__ void operator()(uint32_t k, Fp numer, const EF &den) const {
        const Fp p = numer * norm;  // base-field numerator: limbs 1..3 stay zero
        uint64_t len = 1ull << leaf_log_len[k];
        for (uint64_t off = 0; off < len; off += 1ull << log_height) {
            const uint64_t i = leaf_base[k] + off + row;
            leaves.P()[i] = p.v;  // limbs 1..3 are zero: written by the fill, or never read (p_base)
            stp(leaves.Q(), leaves.stride, i, den);
        }
    }
};
// Prog::operator()(const BaseFetch &fetch, const LeafEmit &emit) evaluates every interaction of the
// AIR at fetch.row and emits (count, denominator) per interaction.
template <class Prog>
__device__ __forceinline__ void gkr_leaves_body(const Prog &prog, const ColDesc *cols, uint32_t log_height,
                                                const uint64_t *leaf_base, const uint32_t *leaf_log_len, Fp norm,
                                                FracPlanes leaves) {
    uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (1ull << log_height)) return;
    BaseFetch fetch{cols, row};
    LeafEmit emit{leaves, leaf_base, leaf_log_len, log_height, row, norm};
    prog(fetch, emit);
}

}  // namespace ovm
