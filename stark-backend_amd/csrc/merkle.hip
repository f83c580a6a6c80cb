// Poseidon2 Merkle commitment (MMCS) over a column-major codeword in PHYSICAL (bit-reversed) row
// order, plus the proof-of-work grind and opening gathers.
//
// Replaces (reference): crates/cuda-backend/cuda/src/merkle_tree.cu:16-79,81-149 (row hashes),
// :153-177 (compress layers), :185-210 (query digest gather), src/sponge.cu:65-117 (grind),
// src/matrix.cu:62-80 (row gather); tree semantics from
// crates/stark-backend/src/prover/stacked_pcs.rs:413-485.
//
// Index spaces. The reference tree is defined on natural evaluation indices m; we store row
// p = bitrev_L(m). With rows_per_query = 2^k and query_stride = 2^(L-k):
//   * the 2^k leaves {i + t*query_stride} of query i are physical rows g*2^k + bitrev_k(t) with
//     g = bitrev_{L-k}(i): one contiguous group;
//   * every layer of the reference tree -- the k strided ones and the adjacent-pair ones above
//     them -- becomes  out[q] = compress(in[q], in[q + half])  on bit-reversed indices, within a
//     group for the first k layers and over the whole layer above. All accesses are unit-stride.
// Stored layer j (j = 0 is the reference's digest_layers[0]) holds entry y at index
// bitrev_{L-k-j}(y).
//
// Roofline: Poseidon2 is integer-ALU bound (~640 32-bit multiplies per permutation), not HBM
// bound; one lane hashes one row and the codeword is read exactly once with unit stride across
// the wavefront.
#include <cstdlib>
#include <cstring>

#include "device_common.h"
#include "device_ctx.h"
#include "openvm_hip.h"
#include "poseidon2.h"
#include "tuning.h"

using bb::Fp;
using bb::FpExt;
using p2::Digest;

namespace ovm {

static constexpr int HASH_THREADS = 256;
#ifndef OVM_HASH_ROWS_WAVES
#define OVM_HASH_ROWS_WAVES  // (", 8": cap the row hash at 64 VGPRs -- measured, see DESIGN)
#endif

// Which rows of the matrix a launch hashes: all of them (n = 1), or -- one rank's share of a tree that n ranks
// build together (prover.hip: build_tree) -- the query groups g with g mod n == r, compacted: output row
// (c << log_group) | t is matrix row (((c * n + r) << log_group) | t).
struct RowMap {
    uint32_t n = 1, r = 0, log_group = 0;
};

// One lane = one row. `words_per_elem` is 1 for base-field matrices and 4 for extension-field
// matrices (stored as [u32;4] per entry); the row is absorbed coefficient-wise.
template <int WORDS>
__global__ __launch_bounds__(HASH_THREADS OVM_HASH_ROWS_WAVES) void hash_rows_kernel(const Fp *__restrict__ mat,
                                                                 uint64_t width, uint64_t rows,
                                                                 uint64_t col_stride,
                                                                 Digest *__restrict__ out, RowMap map) {
    const uint64_t out_row = (uint64_t)blockIdx.x * HASH_THREADS + threadIdx.x;
    if (out_row >= rows) return;
    // (a rank of a distributed tree hashes every dist_n-th query group: map.n == 1 is the identity)
    const uint64_t row = map.n == 1 ? out_row
                                    : ((((out_row >> map.log_group) * map.n + map.r) << map.log_group) | (out_row & ((1ull << map.log_group) - 1)));
    Fp s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = Fp::zero();
    // Absorb 8 words (= CPB columns) per permutation with static register indices. The FIRST permutation of a row sees
    // a zero capacity (and, for a one-element extension-field row, 12 zero words), the LAST one only has to deliver the
    // 8 digest words: p2::permute_t leaves the additions on known zeros / unused words out.
    constexpr int CPB = 8 / WORDS;
    const Fp *src = mat + row * WORDS;
    const uint64_t cstride = col_stride * WORDS;
    auto absorb_full = [&](uint64_t c0) {
#pragma unroll
        for (int j = 0; j < CPB; j++) {
            const Fp *p = src + (c0 + j) * cstride;
            if constexpr (WORDS == 4) {
                uint4 v = *reinterpret_cast<const uint4 *>(p);
                s[4 * j + 0] = Fp::raw(v.x);
                s[4 * j + 1] = Fp::raw(v.y);
                s[4 * j + 2] = Fp::raw(v.z);
                s[4 * j + 3] = Fp::raw(v.w);
            } else {
                s[j] = p[0];
            }
        }
    };
    if (WORDS == 4 && width == 1) {  // the WHIR round codewords: one extension-field element per row, one permutation
        uint4 v = *reinterpret_cast<const uint4 *>(src);
        s[0] = Fp::raw(v.x), s[1] = Fp::raw(v.y), s[2] = Fp::raw(v.z), s[3] = Fp::raw(v.w);
        p2::permute_t<4, true>(s);
    } else {
        // ONE permutation body in the kernel, in a loop that is not unrolled: a copy per position (first / middle / last
        // permutation of the row, with the additions on known zeros and unused words left out) made the kernel 200 KB
        // of straight-line code, three times the instruction cache two CUs share -- and the row hash stopped following
        // the instruction count (round 6: 3 773 -> 3 676 VALU per permutation, 14.5 ms either way).
        const uint64_t chunks = (width + CPB - 1) / CPB;
#pragma nounroll
        for (uint64_t c = 0; c < chunks; c++) {
            const uint64_t c0 = c * CPB;
            if (c0 + CPB <= width) {
                absorb_full(c0);
            } else {
                // partial tail: overwrite the first words only, stale words stay (padding-free sponge)
#pragma unroll
                for (int j = 0; j < CPB; j++) {
                    if (c0 + j < width) {
                        const Fp *p = src + (c0 + j) * cstride;
#pragma unroll
                        for (int k = 0; k < WORDS; k++) s[WORDS * j + k] = p[k];
                    }
                }
            }
            p2::permute(s);
        }
    }
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; i++) d.d[i] = s[i];
    out[out_row] = d;
}

// out[g*half + q] = compress(in[g*2*half + q], in[g*2*half + q + half]),  q < half
__global__ __launch_bounds__(HASH_THREADS) void compress_layer_kernel(const Digest *__restrict__ in,
                                                                      Digest *__restrict__ out,
                                                                      uint64_t n_out,
                                                                      uint32_t log_half) {
    uint64_t o = (uint64_t)blockIdx.x * HASH_THREADS + threadIdx.x;
    if (o >= n_out) return;
    uint64_t q = o & ((1ull << log_half) - 1);
    uint64_t g = o >> log_half;
    uint64_t left = (g << (log_half + 1)) | q;
    Digest l = in[left], r = in[left + (1ull << log_half)];
    out[o] = p2::compress(l, r);
}

// The same with 16 lanes per node (p2::permute_coop): lane i < 8 holds word i of the left digest, lane 8 + i word i
// of the right one. For layers of at most COOP_MAX_NODES nodes: a one-lane permutation is ~12 us of dependent work
// whether the launch has one node or 65 536, the cooperative one ~2 us.
static constexpr uint64_t COOP_MAX_NODES = 1ull << 13;
__global__ __launch_bounds__(HASH_THREADS) void compress_layer_coop_kernel(const Digest *__restrict__ in,
                                                                           Digest *__restrict__ out, uint64_t n_out,
                                                                           uint32_t log_half) {
    const uint64_t t = (uint64_t)blockIdx.x * HASH_THREADS + threadIdx.x;
    const uint64_t o0 = t >> 4;
    const uint32_t lane16 = (uint32_t)(t & 15);
    const uint64_t o = o0 < n_out ? o0 : n_out - 1;  // (whole rows of 16 lanes stay in step for the DPP moves)
    const uint64_t q = o & ((1ull << log_half) - 1), g = o >> log_half;
    const uint64_t left = (g << (log_half + 1)) | q;
    const Digest *src = lane16 < 8 ? in + left : in + left + (1ull << log_half);
    Fp x = src->d[lane16 & 7];
    x = p2::permute_coop(x, lane16);
    if (o0 < n_out && lane16 < 8) out[o].d[lane16] = x;
}

// Proof-of-work search: witness w = base + global thread id; keeps the minimum hit.
__global__ __launch_bounds__(HASH_THREADS) void grind_kernel(p2::Sponge sponge, uint32_t bits,
                                                             uint32_t base, uint32_t count,
                                                             uint32_t *result) {
    uint32_t i = blockIdx.x * HASH_THREADS + threadIdx.x;
    if (i >= count) return;
    uint32_t w = base + i;
    // a smaller witness has been found already: nothing this lane could add (workgroups start roughly in
    // ascending order, so the windows after the first hit cost next to nothing)
    if (*(volatile uint32_t *)result < w) return;
    p2::Sponge s = sponge;
    s.observe(Fp::from_canonical(w));
    if (s.sample_bits(bits) == 0) atomicMin(result, w);
}
// Hands a batch's result to the host through a host-mapped word + sequence flag and re-arms the result word for the next
// batch: no upload of the initial value, no download, no stream synchronisation per batch (6 grinds per config D proof,
// each used to cost two runtime copies and a synchronisation). Its own one-lane launch behind the search: a "last
// workgroup publishes" counter in the search kernel itself cost 2 ms per proof (65 536 workgroups on one address).
__global__ void grind_publish_kernel(uint32_t *result, uint32_t *host_page /* [0] flag, [1] result */, uint32_t seq) {
    const uint32_t found = *result;
    *result = 0xffffffffu;
    st_word_sys(host_page + 1, found);
    release_to_host();
    st_word_sys(host_page, seq);
}

__global__ void permute_states_kernel(Fp *states, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = states[i * 16 + k];
    p2::permute(s);
#pragma unroll
    for (int k = 0; k < 16; k++) states[i * 16 + k] = s[k];
}

__global__ void permute_states_rc_kernel(Fp *states, uint64_t n, const uint32_t *rc) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = states[i * 16 + k];
    p2::permute_rc(s, rc);
#pragma unroll
    for (int k = 0; k < 16; k++) states[i * 16 + k] = s[k];
}

// Gather `n_rows` rows (given by physical row index) of a column-major matrix into a row-major
// buffer of n_rows x width entries.
template <int WORDS>
__global__ void gather_rows_kernel(const Fp *__restrict__ mat, uint64_t width, uint64_t col_stride,
                                   const uint64_t *__restrict__ row_idx, uint64_t n_rows,
                                   Fp *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * width) return;
    uint64_t r = i / width, cidx = i % width;
    const Fp *src = mat + (cidx * col_stride + row_idx[r]) * WORDS;
#pragma unroll
    for (int k = 0; k < WORDS; k++) out[i * WORDS + k] = src[k];
}

__global__ void gather_digests_kernel(const Digest *__restrict__ layers,
                                      const uint64_t *__restrict__ idx, uint64_t n,
                                      Digest *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = layers[idx[i]];
}

// (OVM_MERKLE_COOP=0 keeps every layer on the one-lane kernel: the parity tests compare the two)
static void launch_compress(const Digest *in, Digest *out, uint64_t n_out, uint32_t log_half, hipStream_t stream) {
    const bool coop = tuning().merkle_coop;
    if (coop && n_out <= COOP_MAX_NODES)
        OVM_LAUNCH(compress_layer_coop_kernel, dim3(ceil_div(n_out * 16, HASH_THREADS)), dim3(HASH_THREADS), 0, stream, in,
                           out, n_out, log_half);
    else
        OVM_LAUNCH(compress_layer_kernel, dim3(ceil_div(n_out, HASH_THREADS)), dim3(HASH_THREADS), 0, stream, in, out,
                           n_out, log_half);
}

// log_rows: rows THIS launch hashes (= all rows of the matrix / map.n); the matrix's columns are rows * map.n apart
template <int WORDS>
static int merkle_tree_impl(const uint32_t *d_matrix, uint64_t width, uint32_t log_rows,
                            uint32_t log_rows_per_query, uint32_t *d_scratch,
                            uint32_t *d_digest_layers, hipStream_t stream, RowMap map = RowMap()) {
    if (log_rows_per_query > log_rows || log_rows > 30) return (int)hipErrorInvalidValue;
    const uint64_t rows = 1ull << log_rows;
    map.log_group = log_rows_per_query;
    Digest *leaf = (Digest *)d_scratch;          // rows digests
    Digest *pong = leaf + rows;                  // rows/2 digests (only if k >= 2)
    Digest *layers = (Digest *)d_digest_layers;  // 2*query_stride - 1 digests
    {
        // (rows of at most 8 words are one permutation each: the 1-column common main of a sharded proof --
        // kept apart so that "merkle.hash_rows" stays the wide codeword hash the roofline is quoted on)
        OVM_PROF(WORDS != 1 ? "merkle.hash_rows_ext" : width <= 8 ? "merkle.hash_rows_narrow" : "merkle.hash_rows", stream);
        OVM_LAUNCH(hash_rows_kernel<WORDS>, dim3(ceil_div(rows, HASH_THREADS)),
                           dim3(HASH_THREADS), 0, stream, (const Fp *)d_matrix, width, rows, rows * map.n,
                           log_rows_per_query == 0 ? layers : leaf, map);
        OVM_CHECK_LAUNCH();
    }
    OVM_PROF("merkle.compress_layers", stream);
    // k group-local layers: leaf -> ... -> layers[0]
    const Digest *cur = leaf;
    uint64_t n = rows;
    for (uint32_t j = 0; j < log_rows_per_query; j++) {
        uint64_t n_out = n >> 1;
        bool last = (j + 1 == log_rows_per_query);
        Digest *dst = last ? layers : ((cur == leaf) ? pong : leaf);
        uint32_t log_half = log_rows_per_query - 1 - j;
        launch_compress(cur, dst, n_out, log_half, stream);
        OVM_CHECK_LAUNCH();
        cur = dst;
        n = n_out;
    }
    // global layers
    Digest *layer = layers;
    uint32_t log_n = log_rows - log_rows_per_query;
    while (log_n > 0) {
        uint64_t n_out = 1ull << (log_n - 1);
        Digest *dst = layer + (1ull << log_n);
        launch_compress(layer, dst, n_out, log_n - 1, stream);
        OVM_CHECK_LAUNCH();
        layer = dst;
        log_n--;
    }
    return 0;
}



// One rank's share of a tree that dist_n ranks (a power of two) build together: the query groups g with
// g mod dist_n == dist_r of a matrix of 2^(log_rows_share) * dist_n rows, as a tree of its own (the same buffers as
// ovm_hip_merkle_tree for 2^log_rows_share rows). Its last digest is the root of the subtree whose leaves those
// groups are -- entry dist_r of the full tree's layer of dist_n entries (physical order; see the header comment).
int merkle_tree_share(const uint32_t *d_matrix, uint64_t width, uint32_t log_rows_share, uint32_t log_rows_per_query,
                      uint32_t *d_scratch, uint32_t *d_digest_layers, hipStream_t stream, bool ext, uint32_t dist_n,
                      uint32_t dist_r) {
    if (dist_n == 0 || (dist_n & (dist_n - 1)) || dist_r >= dist_n) return (int)hipErrorInvalidValue;
    RowMap map;
    map.n = dist_n;
    map.r = dist_r;
    if (ext) return merkle_tree_impl<4>(d_matrix, width, log_rows_share, log_rows_per_query, d_scratch, d_digest_layers, stream, map);
    return merkle_tree_impl<1>(d_matrix, width, log_rows_share, log_rows_per_query, d_scratch, d_digest_layers, stream, map);
}

}  // namespace ovm

using namespace ovm;

extern "C" {

uint64_t ovm_hip_merkle_scratch_words(uint32_t log_rows, uint32_t log_rows_per_query) {
    uint64_t rows = 1ull << log_rows;
    uint64_t digests = log_rows_per_query == 0 ? 0 : rows + (rows >> 1);
    return digests * 8;
}
uint64_t ovm_hip_merkle_layers_words(uint32_t log_rows, uint32_t log_rows_per_query) {
    uint64_t qs = 1ull << (log_rows - log_rows_per_query);
    return (2 * qs - 1) * 8;
}

int ovm_hip_merkle_tree(const uint32_t *d_matrix, uint64_t width, uint32_t log_rows,
                        uint32_t log_rows_per_query, uint32_t *d_scratch, uint32_t *d_digest_layers,
                        ovm_stream_t stream) {
    return merkle_tree_impl<1>(d_matrix, width, log_rows, log_rows_per_query, d_scratch,
                               d_digest_layers, (hipStream_t)stream);
}
int ovm_hip_merkle_tree_ext(const uint32_t *d_matrix, uint64_t width, uint32_t log_rows,
                            uint32_t log_rows_per_query, uint32_t *d_scratch,
                            uint32_t *d_digest_layers, ovm_stream_t stream) {
    return merkle_tree_impl<4>(d_matrix, width, log_rows, log_rows_per_query, d_scratch,
                               d_digest_layers, (hipStream_t)stream);
}

int ovm_hip_poseidon2_permute(uint32_t *d_states, uint64_t n, ovm_stream_t stream) {
    if (n == 0) return 0;
    OVM_LAUNCH(permute_states_kernel, dim3(ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, (Fp *)d_states, n);
    OVM_CHECK_LAUNCH();
    return 0;
}

int ovm_hip_poseidon2_permute_rc(uint32_t *d_states, uint64_t n, const uint32_t *d_rc141, ovm_stream_t stream) {
    if (n == 0) return 0;
    if (!d_states || !d_rc141) return (int)hipErrorInvalidValue;
    OVM_LAUNCH(permute_states_rc_kernel, dim3(ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, (Fp *)d_states, n, d_rc141);
    OVM_CHECK_LAUNCH();
    return 0;
}

int ovm_hip_sponge_grind(const ovm_sponge_t *sponge, uint32_t bits, uint32_t *witness_out,
                         ovm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!sponge || !witness_out || bits >= 31) return (int)hipErrorInvalidValue;
    if (bits == 0) {
        *witness_out = 0;
        return 0;
    }
    p2::Sponge s;
    for (int i = 0; i < 16; i++) s.state[i] = Fp::raw(sponge->state[i]);
    s.absorb_idx = sponge->absorb_idx;
    s.sample_idx = sponge->sample_idx;
    if (s.absorb_idx >= 8 || s.sample_idx > 8) return (int)hipErrorInvalidValue;
    OVM_PROF("grind", stream);
    // per (thread, device): the device result word and a host-mapped page (flag, result)
    struct GrindScratch : ovm::ThreadResource {
        uint32_t *p[16] = {}, *h[16] = {}, *hd[16] = {};
        uint32_t seq[16] = {};
        void release() override {
            for (int i = 0; i < 16; i++) {
                if (p[i]) (void)hipFree(p[i]), p[i] = nullptr;
                if (h[i]) (void)hipHostFree(h[i]), h[i] = hd[i] = nullptr;
            }
        }
        ~GrindScratch() override {
            if (!ovm::runtime_exiting()) release();
        }
    };
    static thread_local GrindScratch scratch;
    int dev = 0;
    OVM_CHECK(hipGetDevice(&dev));
    dev &= 15;
    if (!scratch.p[dev]) {
        const uint32_t init[2] = {0xffffffffu, 0};  // (re-armed by every publish)
        OVM_CHECK(hipMalloc(&scratch.p[dev], 64));
        OVM_CHECK(hipMemcpy(scratch.p[dev], init, sizeof init, hipMemcpyHostToDevice));
        void *hp = nullptr, *dp = nullptr;
        OVM_CHECK(hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent));
        OVM_CHECK(hipHostGetDevicePointer(&dp, hp, 0));
        memset(hp, 0, 64);
        scratch.h[dev] = (uint32_t *)hp, scratch.hd[dev] = (uint32_t *)dp;
    }
    volatile uint32_t *page = scratch.h[dev];
    // Expected 2^bits candidates per hit: scan ascending in batches of four times that (a batch holds a
    // hit with probability 1 - e^-4; the candidates above the first hit exit early), so the first batch
    // with a hit yields the global minimum witness.
    uint64_t batch = 4ull << (bits < 16 ? 16 : bits);
    if (batch > (1ull << 24)) batch = 1ull << 24;
    for (uint64_t base = 0; base < bb::P; base += batch) {
        uint32_t count = (uint32_t)((bb::P - base) < batch ? (bb::P - base) : batch);
        const uint32_t seq = ++scratch.seq[dev];
        OVM_LAUNCH(grind_kernel, dim3(ceil_div(count, HASH_THREADS)), dim3(HASH_THREADS), 0,
                           stream, s, bits, (uint32_t)base, count, scratch.p[dev]);
        OVM_LAUNCH(grind_publish_kernel, dim3(1), dim3(1), 0, stream, scratch.p[dev], scratch.hd[dev], seq);
        OVM_CHECK_LAUNCH();
        uint64_t spins = 0;
        while (page[0] != seq) {
            if ((++spins & 0xFFFF) == 0) {  // a faulted kernel never raises the flag
                const hipError_t q = hipStreamQuery(stream);
                if (q != hipSuccess && q != hipErrorNotReady) return (int)q;
                if (q == hipSuccess && page[0] != seq && spins > (1ull << 34)) return (int)hipErrorUnknown;
            }
        }
        const uint32_t found = page[1];
        if (found != 0xffffffffu) {
            *witness_out = found;
            return 0;
        }
    }
    return (int)hipErrorNotFound;
}

int ovm_hip_gather_rows(const uint32_t *d_matrix, uint64_t width, uint64_t col_stride,
                        uint32_t words_per_elem, const uint64_t *d_row_idx, uint64_t n_rows,
                        uint32_t *d_out, ovm_stream_t stream) {
    uint64_t total = n_rows * width;
    if (total == 0) return 0;
    if (words_per_elem == 1)
        OVM_LAUNCH(gather_rows_kernel<1>, dim3(ceil_div(total, 256)), dim3(256), 0,
                           (hipStream_t)stream, (const Fp *)d_matrix, width, col_stride, d_row_idx,
                           n_rows, (Fp *)d_out);
    else if (words_per_elem == 4)
        OVM_LAUNCH(gather_rows_kernel<4>, dim3(ceil_div(total, 256)), dim3(256), 0,
                           (hipStream_t)stream, (const Fp *)d_matrix, width, col_stride, d_row_idx,
                           n_rows, (Fp *)d_out);
    else
        return (int)hipErrorInvalidValue;
    OVM_CHECK_LAUNCH();
    return 0;
}

int ovm_hip_gather_digests(const uint32_t *d_layers, const uint64_t *d_idx, uint64_t n,
                           uint32_t *d_out, ovm_stream_t stream) {
    if (n == 0) return 0;
    OVM_LAUNCH(gather_digests_kernel, dim3(ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const Digest *)d_layers, d_idx, n, (Digest *)d_out);
    OVM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
