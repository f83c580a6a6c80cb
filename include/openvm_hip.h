/*
 * openvm_hip.h -- C ABI of the MI355X (gfx950) prover backend for openvm-org/stark-backend v2
 * (SWIRL over BabyBear + Poseidon2).
 *
 * This is the drop-in boundary: the functions a Rust `openvm-hip-backend` crate binds with
 * `extern "C"` exactly like `openvm-cuda-backend` binds its kernels
 * (reference: the files under crates/cuda-backend/src/cuda/). Conventions follow the reference FFI:
 *   - every entry returns `int`: 0 on success, otherwise a `hipError_t` value
 *     (reference: crates/cuda-common/include/launcher.cuh:41-54, crates/cuda-common/src/error.rs:54-60);
 *     invalid arguments return hipErrorInvalidValue (= 1), cf. merkle_tree.cu:220, supra/ntt.cu:215;
 *   - field elements cross as raw 32-bit MONTGOMERY words (a Rust `BabyBear` is a
 *     `repr(transparent)` Montgomery u32, an extension element is `[u32; 4]`, 4-byte aligned);
 *   - `d_` pointers are device pointers, everything else is host memory;
 *   - work is enqueued on the given stream; entries that return host values synchronise it.
 *
 * Device matrices are column-major planes (row index fastest), as in the reference
 * (crates/cuda-backend/src/base.rs:8-12). Codewords and Merkle layers are kept in
 * bit-reversed ("physical") row order; see the per-function notes and DESIGN.md.
 */
#ifndef OPENVM_HIP_H
#define OPENVM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without requiring the HIP headers. NULL = default stream. */
typedef void *ovm_stream_t;

/* Fiat-Shamir duplex sponge state (Poseidon2 width 16, rate 8, overwrite mode).
 * Replaces `DeviceSpongeState` (reference: crates/cuda-backend/src/sponge.rs:39-48). */
typedef struct ovm_sponge_t {
    uint32_t state[16];  /* Montgomery words */
    uint32_t absorb_idx; /* 0 <= absorb_idx < 8 */
    uint32_t sample_idx; /* 0 <= sample_idx <= 8 */
} ovm_sponge_t;

/* ---------------------------------------------------------------------------------------
 * Library / device
 * ------------------------------------------------------------------------------------- */

/* ABI version of this header. */
uint32_t ovm_hip_abi_version(void);
/* Selects the HIP device for the calling thread and builds the twiddle tables.
 * Replaces `_generate_all_twiddles` / `_generate_partial_twiddles` / `_generate_device_ntt_twiddles`
 * (reference: crates/cuda-backend/src/cuda/ntt.rs:12-22, batch_ntt_small.rs:49) and
 * `_cuda_get_sm_count` (device_info.rs:4): *cu_count receives the number of compute units. */
int ovm_hip_init(int device, uint32_t *cu_count);
/* Releases the calling thread's device resources NOW -- result mailboxes, the pinned upload arena, the proof-of-work
 * scratch word, the free segments of its memory pools -- instead of at thread exit; for hosts that tear the HIP
 * runtime down themselves or recycle worker threads. They are created again on demand. (At process exit the
 * library's thread-exit destructors leave device memory alone rather than call into a runtime that is going away.) */
int ovm_hip_thread_shutdown(void);
/* Name of a non-zero return code (hipGetErrorName; OVM_ESHARD_* for the exchange's own codes). */
const char *ovm_hip_error_name(int code);

/* Per-stage timing with HIP events on the launching stream (reference: the `tracing` spans /
 * NVTX ranges of crates/stark-sdk/src/{metrics_tracing,nvtx_tracing}.rs). enable(1) resets and
 * starts recording every stage, enable(2) records only the dominant kernel's label
 * ("merkle.hash_rows": one event pair per commitment -- what a timed benchmark region can afford),
 * enable(0) stops. report() synchronises the device and writes one "label calls total_ms" line per
 * label; returns the buffer size needed. With OVM_ROCTX=1 in the environment the same labels are also
 * emitted as roctx ranges (rocprofv3 --marker-trace), the analogue of nvtx_tracing.rs:9-33. */
int ovm_hip_profile_enable(int on);
uint64_t ovm_hip_profile_report(char *buf, uint64_t cap);

/* Plain device memory management for callers without a HIP binding of their own
 * (reference: crates/cuda-common/src/d_buffer.rs, copy.rs). */
int ovm_hip_malloc(void **d_ptr, size_t bytes);
int ovm_hip_free(void *d_ptr);
int ovm_hip_memcpy_h2d(void *d_dst, const void *src, size_t bytes, ovm_stream_t stream);
int ovm_hip_memcpy_d2h(void *dst, const void *d_src, size_t bytes, ovm_stream_t stream);
int ovm_hip_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, ovm_stream_t stream);
int ovm_hip_memset(void *d_dst, int value, size_t bytes, ovm_stream_t stream);
int ovm_hip_stream_create(ovm_stream_t *stream);
int ovm_hip_stream_destroy(ovm_stream_t stream);
int ovm_hip_stream_synchronize(ovm_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * NTT / Reed-Solomon encode
 * ------------------------------------------------------------------------------------- */

/* Per column and per 2^l_skip-row chunk: inverse DFT over the chunk, then the subset-zeta
 * transform over the chunk's index bits ("eval-to-coeff RS message",
 * reference: crates/stark-backend/src/prover/poly.rs:325-348). d_in/d_out: width columns of
 * 2^log_height rows; may alias. Replaces `_batch_ntt_small` + `_mle_interpolate_stage*`
 * as used by crates/cuda-backend/src/stacked_pcs.rs:262-308. */
int ovm_hip_eval_to_coeff(const uint32_t *d_in, uint32_t *d_out, uint64_t width,
                          uint32_t log_height, uint32_t l_skip, ovm_stream_t stream);

/* Decimation-in-frequency NTT of `width` columns: input rows [0, 2^log_n_in) in natural order
 * (rows above that are taken as zero), output 2^log_n rows in BIT-REVERSED order:
 *   d_out[col * 2^log_n + p] = sum_i in[col][i] * w^(i * bitrev_{log_n}(p)),  w = two_adic_generator(log_n).
 * d_in columns are 2^log_n_in apart, d_out columns 2^log_n apart. In-place (d_in == d_out) is
 * allowed when log_n_in == log_n. Replaces `_ct_mixed_radix_narrow` + `_bit_rev`
 * (reference: crates/cuda-backend/src/cuda/ntt.rs:50-88,161-171). */
int ovm_hip_ntt_dif(const uint32_t *d_in, uint32_t *d_out, uint64_t width, uint32_t log_n,
                    uint32_t log_n_in, ovm_stream_t stream);
/* Same for extension-field columns ([u32;4] entries, base-field twiddles). */
int ovm_hip_ntt_dif_ext(const uint32_t *d_in, uint32_t *d_out, uint64_t width, uint32_t log_n,
                        uint32_t log_n_in, ovm_stream_t stream);

/* Reed-Solomon encode of a stacked matrix (`rs_code_matrix`, reference:
 * crates/stark-backend/src/prover/stacked_pcs.rs:341-367, crates/cuda-backend/src/stacked_pcs.rs:229-337):
 * codeword column = DFT over the subgroup of order 2^(log_height+log_blowup) of the zero-padded
 * eval-to-coeff message, stored in bit-reversed row order (physical row p holds natural index
 * bitrev(p)). d_scratch: width * 2^log_height words (may alias d_stacked if the stacked
 * evaluations are no longer needed). d_codeword: width * 2^(log_height+log_blowup) words. */
int ovm_hip_rs_encode(const uint32_t *d_stacked, uint32_t *d_scratch, uint32_t *d_codeword,
                      uint64_t width, uint32_t log_height, uint32_t l_skip, uint32_t log_blowup,
                      ovm_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Poseidon2 Merkle tree and transcript proof-of-work (row A5)
 * ------------------------------------------------------------------------------------- */

/* Sizes, in 32-bit words, of the buffers ovm_hip_merkle_tree needs. */
uint64_t ovm_hip_merkle_scratch_words(uint32_t log_rows, uint32_t log_rows_per_query);
uint64_t ovm_hip_merkle_layers_words(uint32_t log_rows, uint32_t log_rows_per_query);

/* Merkle tree of `MerkleTree::new` (reference: crates/stark-backend/src/prover/stacked_pcs.rs:413-485)
 * over a column-major matrix of 2^log_rows rows given in bit-reversed row order.
 * d_digest_layers receives the stored layers back to back (2^(log_rows-k) + ... + 1 digests of
 * 8 words); layer j holds the reference's digest_layers[j][y] at index bitrev_{log_rows-k-j}(y).
 * The last digest is the root. Replaces `_poseidon2_compressing_row_hashes[_ext]`,
 * `_poseidon2_strided_compress_layer`, `_poseidon2_adjacent_compress_layer`
 * (reference: crates/cuda-backend/src/cuda/merkle_tree.rs:18-48). */
int ovm_hip_merkle_tree(const uint32_t *d_matrix, uint64_t width, uint32_t log_rows,
                        uint32_t log_rows_per_query, uint32_t *d_scratch, uint32_t *d_digest_layers,
                        ovm_stream_t stream);
/* Same for an extension-field matrix (rows absorbed coefficient-wise). */
int ovm_hip_merkle_tree_ext(const uint32_t *d_matrix, uint64_t width, uint32_t log_rows,
                            uint32_t log_rows_per_query, uint32_t *d_scratch,
                            uint32_t *d_digest_layers, ovm_stream_t stream);

/* n independent Poseidon2 permutations of 16-word states, in place (test / micro-benchmark entry;
 * reference: crates/cuda-common/include/poseidon2.cuh:184-202). */
int ovm_hip_poseidon2_permute(uint32_t *d_states, uint64_t n, ovm_stream_t stream);
/* The same S-box and linear layers over round constants read from device memory: d_rc141 = 64 initial, 64
 * terminal, 13 internal Montgomery words (what `Poseidon2BabyBear::new_from_rng_128` would build, p3-baby-bear
 * 0.4.3). Test entry: runs Plonky3's published known-answer vector, whose constants are drawn from a seeded RNG,
 * through the product's own layer code (tests/golden/poseidon2_p3_kat.json). */
int ovm_hip_poseidon2_permute_rc(uint32_t *d_states, uint64_t n, const uint32_t *d_rc141, ovm_stream_t stream);

/* Proof-of-work: the MINIMUM canonical witness w such that observing w and sampling yields
 * `bits` zero low bits (reference: crates/stark-backend/src/transcript/traits.rs:62-89; replaces
 * `_sponge_grind`, crates/cuda-backend/src/cuda/sponge.rs:13). `sponge` is not modified.
 * *witness_out is the canonical integer value. Synchronises the stream. */
int ovm_hip_sponge_grind(const ovm_sponge_t *sponge, uint32_t bits, uint32_t *witness_out,
                         ovm_stream_t stream);

/* Row gather for query openings: out[r * width + c] = matrix[c][d_row_idx[r]]
 * (replaces `_matrix_get_rows_fp`, reference: crates/cuda-backend/src/cuda/matrix.rs:24-32). */
int ovm_hip_gather_rows(const uint32_t *d_matrix, uint64_t width, uint64_t col_stride,
                        uint32_t words_per_elem, const uint64_t *d_row_idx, uint64_t n_rows,
                        uint32_t *d_out, ovm_stream_t stream);
/* Digest gather for Merkle paths: out[i] = layers[d_idx[i]] (8 words each)
 * (replaces `_query_digest_layers`, reference: crates/cuda-backend/src/cuda/merkle_tree.rs:50-58). */
int ovm_hip_gather_digests(const uint32_t *d_layers, const uint64_t *d_idx, uint64_t n,
                           uint32_t *d_out, ovm_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Proving key and the prove entry point
 * ------------------------------------------------------------------------------------- */

/* Device-side proving key. Replaces `DeviceMultiStarkProvingKey<GpuBackend>` as produced by
 * `transport_pk_to_device` (reference: crates/stark-backend/src/prover/hal.rs:141-160,
 * crates/cuda-backend/src/data_transporter.rs:35-79). */
typedef struct ovm_pk_t ovm_pk_t;
typedef struct ovm_pcs ovm_pcs_t; /* opaque StackedPcsData: stacked matrix, codeword, Merkle layers */

/* Builds a proving key from the flat u32 "proving-key blob" (all field constants CANONICAL):
 *   0x4F564D31, l_skip, n_stack, w_stack, log_blowup, k_whir, mu_pow_bits, query_phase_pow_bits,
 *   folding_pow_bits, logup_pow_bits, max_constraint_degree, num_whir_rounds, num_queries[rounds],
 *   vk_pre_hash[8], num_airs, then per AIR:
 *     need_rot, is_required, max_constraint_degree, num_public_values, has_preprocessed,
 *     preprocessed_width, num_cached, cached_width[num_cached], common_main_width,
 *     num_nodes, nodes[num_nodes] as 6 words (kind, entry, part_index, offset, x, y) with
 *       kind: 0 Variable, 1 IsFirstRow, 2 IsLastRow, 3 IsTransition, 4 Constant, 5 Add, 6 Sub,
 *             7 Neg, 8 Mul (reference: crates/stark-backend/src/air_builders/symbolic/dag.rs:20-47);
 *       entry: 0 Preprocessed, 1 Main, 2 Public (symbolic_variable.rs:13-27);
 *       x = column index | constant value | left node, y = right node;
 *     num_constraints, constraint_idx[], num_interactions, then per interaction
 *     bus_index, count_weight, count_node, message_len, message_nodes[].
 * Main variables: part_index 0..num_cached-1 are the cached main partitions, part_index num_cached
 * is the common main (air_builders/symbolic/mod.rs:117-127).
 * If any AIR has a preprocessed trace the blob ends with: 0x50524550 ('PREP'), then for every such
 * AIR in order: height, preprocessed_width*height canonical values, column-major. The key uploads
 * and commits each of them on `stream` (keygen/mod.rs:283-300, hal.rs:141-150) and synchronises.
 * This is what the Rust side serialises from `MultiStarkProvingKey` (vk + preprocessed traces)
 * plus `vk_pre_hash`. */
int ovm_hip_pk_load(const uint32_t *blob, uint64_t words, ovm_pk_t **pk_out, ovm_stream_t stream);
void ovm_hip_pk_free(ovm_pk_t *pk);
/* Commitment (Montgomery words) of AIR air_idx's preprocessed trace as committed at load time --
 * must equal `vk.preprocessed_data.commit` (keygen/types.rs:99-107). hipErrorInvalidValue if the
 * AIR has none. */
int ovm_hip_pk_preprocessed_commitment(const ovm_pk_t *pk, uint32_t air_idx, uint32_t commitment_out[8]);
/* The key's own device copy of that trace (column-major, `width` x `height` Montgomery words) and its PCS data, BORROWED:
 * valid while the key lives, never passed to ovm_hip_pcs_free. This is what a backend hands out as
 * `DeviceStarkProvingKey::preprocessed_data` (prover/types.rs:18-25 CommittedTraceData{commitment, trace, data};
 * CUDA peer: data_transporter.rs:44-51) without uploading and committing the trace a second time. */
int ovm_hip_pk_preprocessed_data(const ovm_pk_t *pk, uint32_t air_idx, const uint32_t **d_trace_out, uint64_t *width_out,
                                 uint64_t *height_out, uint32_t commitment_out[8], ovm_pcs_t **pcs_out);
/* Host-only self-test of the forms the kernels execute (csrc/program.h: stage_cached + linear_terms for round 0 and
 * the GKR leaves, compile_grouped + encode_mle_words for the multilinear rounds): every program of every AIR of the
 * key, laid out with `stage_rows` / `acc_rows` rows, is evaluated on one random row next to a direct evaluation of
 * the DAG. Needs no device. 0 = all equal; `blocks_out` (optional) = blocks staged. */
int ovm_hip_staged_selftest(const uint32_t *blob, uint64_t words, uint32_t stage_rows, uint32_t acc_rows,
                            uint64_t seed, uint64_t *blocks_out);
/* Host-only: size of the register program one AIR of the key compiles to (kind 0 zerocheck, 1 LogUp round sums,
 * 2 GKR leaves): instructions, slots (simultaneously live intermediates) and the recomputation limit the compiler
 * settled on (csrc/program.h compile_program; 0 = nothing is computed twice). Diagnostics / tests. */
int ovm_hip_program_stats(const uint32_t *blob, uint64_t words, uint32_t air_idx, uint32_t kind, uint32_t *instrs_out,
                          uint32_t *slots_out, uint32_t *remat_limit_out);
/* Host-only self-test of the transcript's permutation: the run-time selected host implementation
 * (AVX2 where the CPU has it, csrc/host_poseidon2.cpp) against the scalar one on n pseudo-random
 * states. 1 = equal on all of them. */
int ovm_hip_host_permute_selftest(uint32_t n, uint64_t seed);
/* Host-only self-test of the vectorised host arithmetic of the LogUp-GKR layer tails (csrc/host_gkr_batch.cpp: AVX2 where
 * the CPU has it) against the scalar code on `rounds` x 40 random batches. 1 = equal on all of them. */
int ovm_hip_host_gkr_batch_selftest(uint32_t rounds, uint64_t seed);
/* Memory-bounded modes of the calling thread's provers (reference: GpuProverConfig{cache_stacked_matrix,
 * cache_rs_code_matrix, zerocheck_save_memory}, crates/cuda-backend/src/device.rs:25-30). Defaults: everything
 * resident (all fields zero except cache_rs_code_matrix = 1).
 *   cache_rs_code_matrix = 0   a commitment made from now on drops its codeword once the Merkle tree is built
 *                              and re-encodes it when WHIR opens rows of it (reference: whir.rs:404-416): -3.4 GB
 *                              per 2^22 x 100 commitment at blowup 2, one extra NTT at opening time.
 *   zerocheck_save_memory = 1  round 0 extends and evaluates a tall trace in slabs of 2^round0_slab_log_rows rows
 *                              (0 = default 20): the coset-LDE buffer shrinks from
 *                              ovm_hip_constraint_eval_buffer_size(pk, air) x height to that x the slab. (The slab
 *                              size is a lower bound: a proof whose traces would need more than ~64 K result
 *                              slots over all slabs -- hundreds of slabs per trace -- gets larger slabs.)
 *   gkr_drop_leaves = 1        the leaf layer of the LogUp fraction tree (the largest: 5 of the tree's 13 units)
 *                              is dropped once the layer above it is built and generated again, from the trace,
 *                              when the last layer's sumcheck needs it -- by then the rest of the tree is gone.
 *                              Peak of the tree: leaves + first layer (9 units) instead of everything (13). The
 *                              reference's CUDA backend halves its tree by un-adding fractions in place
 *                              (docs/cuda-backend/gkr-prover.md:129-133), which costs extension-field inversions;
 *                              generating the leaves twice costs 0.7 ms at config D.
 *   cache_limit_bytes          upper bound on the device memory the thread's caching allocator keeps in free
 *                              blocks (0 = unbounded); blocks beyond it go back to the driver.
 * Proofs are identical in every mode. The stacked matrix is never copied when one trace fills it
 * (cache_stacked_matrix has no counterpart here). */
typedef struct ovm_prover_config {
    uint32_t cache_rs_code_matrix;
    uint32_t zerocheck_save_memory;
    uint32_t round0_slab_log_rows;
    uint32_t gkr_drop_leaves;
    uint64_t cache_limit_bytes;
} ovm_prover_config_t;
int ovm_hip_set_prover_config(const ovm_prover_config_t *cfg);
/* Device memory handed out by the library's allocator, process-wide: now, at its peak (reset_peak != 0 restarts
 * the peak from the current value), and the calling thread's cached free blocks. Any pointer may be null. */
int ovm_hip_memory_stats(uint64_t *live_bytes, uint64_t *peak_live_bytes, uint64_t *cached_bytes, int reset_peak);
/* The pool behind it (csrc/block_pool.h; the role of the reference's VPMM pool, docs/vpmm_spec.md), process-wide:
 * out4 = { bytes held from the driver in segments now, their peak (restarted by reset_peak above), hipMalloc calls
 * ever made, blocks handed out now }. Back-to-back proofs of one shape make no driver call after the first. */
int ovm_hip_pool_stats(uint64_t out4[4]);
/* Device-free self-test of that pool over host memory: `ops` random allocations / frees of both size classes with
 * the invariants checked (no overlap, contents intact, accounting, limit, a repeated sequence makes no driver call,
 * nothing left after trim). 0 = all held; message: ovm_hip_selftest_error(). */
int ovm_hip_pool_selftest(uint64_t seed, uint32_t ops);
/* ProverBackend::constraint_eval_buffer_size (hal.rs:57-62): bytes of round-0 scratch per trace ROW of this AIR
 * (the coset LDE of every view column on max(degree - 1, degree if it has interactions) cosets); what memory
 * metering multiplies by the trace height -- or by the slab with zerocheck_save_memory. 0 for a bad index. */
uint64_t ovm_hip_constraint_eval_buffer_size(const ovm_pk_t *pk, uint32_t air_idx);
/* Returns the calling thread's cached device blocks to the driver (the prover recycles its
 * buffers between proofs; reference: VPMM pool, crates/cuda-common/src/memory_manager/). */
int ovm_hip_trim(void);
/* Message of the last failing call on this thread (empty if none). */
const char *ovm_hip_last_error(void);

/* ---- stage-level entries: the three backend traits of crates/stark-backend/src/prover/hal.rs ----
 * A Rust `HipDevice` implements TraceCommitter / MultiRapProver / OpeningProver with one call each
 * and leaves `Coordinator::prove` (prover/mod.rs:104-197), the proof header and the transcript
 * observations between the stages to the reference's own code. The sponge is the host twin of the
 * transcript state (reference: DeviceSpongeState, crates/cuda-backend/src/sponge.rs:39-48), read
 * at entry and written back at exit. Traces may be passed in any order; the stages sort them as
 * `ProvingContext::into_sorted` does (prover/types.rs:137-148). */

/* One AIR's proving input (reference: AirProvingContext, prover/types.rs:59-73). POD. */
typedef struct ovm_trace {
    uint32_t air_idx;               /* position in the key's AIR list */
    uint32_t num_public_values;
    const uint32_t *d_common_main;  /* device, column-major Montgomery words: height x common_main_width */
    uint64_t height;                /* power of two */
    const uint32_t *public_values;  /* host, Montgomery words */
    uint32_t num_cached;            /* = the AIR's number of cached main partitions */
    uint32_t _pad;
    ovm_pcs_t *const *cached;       /* host array of handles from ovm_hip_commit_matrix, partition order */
} ovm_trace_t;

/* TraceCommitter::commit (hal.rs:84-87; CUDA backend: crates/cuda-backend/src/stacked_pcs.rs:
 * 229-337 + merkle_tree.rs) of the common mains of `traces`: stack, eval->coeff, RS encode, Merkle
 * tree. commitment_out = root (Montgomery words). When one trace fills the stacked matrix exactly
 * the PCS data aliases the trace buffer instead of copying it: keep the traces alive until
 * prove_openings has returned (the reference's ctx owns them for exactly that long,
 * hal.rs:123-137). Synchronises the stream. */
int ovm_hip_commit(const ovm_pk_t *pk, uint32_t n_traces, const ovm_trace_t *traces, uint32_t commitment_out[8],
                   ovm_pcs_t **pcs_out, ovm_stream_t stream);
/* The same for ONE matrix of any width: how a cached main partition is committed before proving
 * (reference callers: `stacked_commit(.., &[&cached])`, test_utils/mod.rs:297-318; the result is a
 * CommittedTraceData, prover/types.rs:18-25). The handle keeps pointing at d_matrix: keep it alive
 * as long as the handle. */
int ovm_hip_commit_matrix(const ovm_pk_t *pk, const uint32_t *d_matrix, uint64_t width, uint64_t height,
                          uint32_t commitment_out[8], ovm_pcs_t **pcs_out, ovm_stream_t stream);
/* TraceCommitter::commit exactly as the trait has it (hal.rs:84-87: `commit(&self, &[&Matrix])`, no key,
 * no AIR indices; CUDA peer: `stacked_commit(l_skip, n_stack, log_blowup, k_whir, traces, ..)`,
 * crates/cuda-backend/src/stacked_pcs.rs:50-88): stacks the n matrices, which MUST be sorted by height
 * descending (stacked_pcs.rs:294), eval->coeff, RS encode, Merkle tree. The handle points into the
 * matrices when one of them fills the stacked matrix exactly: keep them alive as long as the handle. */
typedef struct ovm_pcs_params {
    uint32_t l_skip, n_stack, log_blowup, k_whir;
} ovm_pcs_params_t;
typedef struct ovm_matrix {
    const uint32_t *d_ptr; /* device, column-major Montgomery words */
    uint64_t width, height; /* height a power of two */
} ovm_matrix_t;
int ovm_hip_commit_matrices(const ovm_pcs_params_t *params, uint32_t n, const ovm_matrix_t *mats, uint32_t commitment_out[8],
                            ovm_pcs_t **pcs_out, ovm_stream_t stream);
/* Exact size of the proof `ovm_hip_prove` writes for traces of these shapes (only air_idx and height of
 * each entry are read): byte offsets of the ends of the header / GKR / batch-constraint / stacking / WHIR
 * sections; section_ends5[4] is the proof length. The encoding has no data-dependent lengths, so buffers
 * can be sized once and no call ever has to be repeated (reference: the buffer-size queries of
 * crates/cuda-backend/src/cuda/logup_zerocheck.rs:152,344-352). Needs no device. */
int ovm_hip_proof_size(const ovm_pk_t *pk, uint32_t n_traces, const ovm_trace_t *traces, uint64_t section_ends5[5]);
void ovm_hip_pcs_free(ovm_pcs_t *pcs);

/* MultiRapProver::prove_rap_constraints (hal.rs:94-112; CUDA backend: crates/cuda-backend/src/
 * logup_zerocheck/mod.rs): LogUp PoW + GKR, zerocheck/logup batch sumcheck, column openings.
 * out = GkrProof bytes (gkr_len) followed by BatchConstraintProof bytes, each in the encoding of
 * proof.rs:226-421 (`GkrProof::decode` / `BatchConstraintProof::decode` read them).
 * r_out = the sumcheck point (the trait's `Artifacts = Vec<EF>`), 4 Montgomery words per element,
 * r_len elements. Synchronises the stream. If out_cap or r_cap is too small: hipErrorInvalidValue with
 * *total_len / *r_len = the sizes needed and the sponge untouched (sizes beforehand: ovm_hip_proof_size;
 * r_len = l_skip-independent: 1 + max over traces of max(log_height - l_skip, 0)). */
int ovm_hip_prove_rap_constraints(const ovm_pk_t *pk, uint32_t n_traces, const ovm_trace_t *traces,
                                  ovm_sponge_t *sponge, uint8_t *out, uint64_t out_cap, uint64_t *gkr_len,
                                  uint64_t *total_len, uint32_t *r_out, uint32_t r_cap, uint32_t *r_len,
                                  ovm_stream_t stream);

/* OpeningProver::prove_openings (hal.rs:118-138; CUDA backend: stacked_reduction.rs + whir.rs):
 * stacked opening reduction and WHIR over all commitments -- the common main, then per trace its
 * preprocessed commitment (from the key) and its cached mains (cpu-backend/src/device.rs:147-172).
 * out = StackingProof bytes (stacking_len) followed by WhirProof bytes. d_common_main and
 * public_values of `traces` are not read. Synchronises the stream. Too small an `out`: as above (sponge
 * untouched, *total_len = the size needed). */
int ovm_hip_prove_openings(const ovm_pk_t *pk, uint32_t n_traces, const ovm_trace_t *traces, ovm_pcs_t *common_main,
                           const uint32_t *r, uint32_t r_len, ovm_sponge_t *sponge, uint8_t *out, uint64_t out_cap,
                           uint64_t *stacking_len, uint64_t *total_len, ovm_stream_t stream);

/* `Coordinator::prove` for traces already resident on the device (reference:
 * crates/stark-backend/src/prover/mod.rs:104-197 driving hal.rs:84-138): commits the common main
 * traces, proves the constraints (LogUp-GKR + batch sumcheck) and the openings (stacked reduction
 * + WHIR), and writes the proof in the reference's canonical byte encoding
 * (crates/stark-backend/src/proof.rs:226-421) -- `Proof::decode_from_bytes` reads it.
 *   sponge: transcript state, updated in place (NULL = fresh transcript).
 *   section_offsets5 (optional): byte offsets of the ends of header / GKR / batch-constraint /
 *   stacking / WHIR sections.
 * PoW witnesses are the minimum valid ones. Synchronises the stream.
 * If proof_cap is too small: returns hipErrorInvalidValue (1) with *proof_len = the size needed, BEFORE
 * doing any work (the proof length depends on the shapes only, see ovm_hip_proof_size); the sponge is
 * untouched. */
int ovm_hip_prove(const ovm_pk_t *pk, uint32_t n_traces, const ovm_trace_t *traces, ovm_sponge_t *sponge,
                  uint8_t *proof_out, uint64_t proof_cap, uint64_t *proof_len, uint64_t *section_offsets5,
                  ovm_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Sharded proving: ONE joint proof over the ranks of one node, one AIR (trace) per GPU
 * (BASELINE config D/E in their 8 x MI355X form). DESIGN.md §6 has the protocol.
 * ------------------------------------------------------------------------------------- */

/* The communicator of a sharded call. The exchanges:
 *   allgather_host   small host buffers, hundreds of times per proof (per-round partial sums of the
 *                    sumchecks, roots, openings): all[r*bytes ..] = rank r's `mine`; every rank calls
 *                    it in the same order with the same `bytes`. Built in: ovm_hip_shm_allgather.
 *   allgather_device device buffers: d_all[r*bytes ..] = rank r's d_mine. Used for each AIR's share of
 *                    the common main at the start and for the mu-batched WHIR vector (reference: the
 *                    batching step of prover/whir.rs:95-132).
 *   alltoall_device  OPTIONAL (may be null): d_recv[s*bytes ..] = rank s's d_send[rank*bytes ..]. With it
 *                    the N per-rank mu-batch vectors are summed as all-to-all + local modular add of
 *                    one slice + all-gather of the slices -- 2*(N-1)/N of ONE vector in and out of every
 *                    GPU, spread over all its xGMI links -- instead of an all-gather of N whole vectors
 *                    ((N-1) vectors into every GPU). Without it the all-gather path runs.
 * The device exchanges are supplied by the host language on top of RCCL over xGMI (torch.distributed in
 * this repository's Python host, an rccl binding in Rust); each must be ordered after the work already
 * enqueued on `stream` and complete, from the point of view of later work on `stream`, when it returns.
 *   abort            OPTIONAL (may be null): called by a rank whose part of a joint call FAILED after the ranks had
 *                    started working (an exception, out of memory, a failing kernel): "I will not make my remaining
 *                    contributions". The built-in exchange raises a flag in the shared segment: peers waiting in (or
 *                    entering) allgather_host return OVM_ESHARD_PEER_ABORTED at once instead of spinning until the
 *                    time-out, and the communicator is POISONED on every rank (later calls return
 *                    OVM_ESHARD_POISONED): after an aborted call the ranks' sequence numbers differ, and reusing the
 *                    communicator could pair a fresh contribution with a stale one. The host destroys and recreates it.
 *                    Input validation is NOT such a failure: ovm_hip_prove_sharded exchanges every rank's validation
 *                    result before any work, all ranks refuse together, and the communicator stays usable.
 * All return 0 or a hipError_t value (or one of the OVM_ESHARD_* codes below). */
typedef struct ovm_shard {
    uint32_t rank, world;
    void *ctx;
    int (*allgather_host)(void *ctx, const void *mine, void *all, uint64_t bytes);
    int (*allgather_device)(void *ctx, const void *d_mine, void *d_all, uint64_t bytes, ovm_stream_t stream);
    int (*alltoall_device)(void *ctx, const void *d_send, void *d_recv, uint64_t bytes, ovm_stream_t stream);
    int (*abort)(void *ctx, int code);
} ovm_shard_t;
/* Return codes of the built-in exchange that are not hipError_t values. */
#define OVM_ESHARD_TIMEOUT 10006      /* a peer did not contribute within OVM_SHARD_TIMEOUT_S seconds (default 300; a peer that FAILS is announced through the abort word within milliseconds) */
#define OVM_ESHARD_PEER_ABORTED 10007 /* a peer called abort() */
#define OVM_ESHARD_POISONED 10008     /* this communicator saw one of the two above earlier */
/* Message of the calling thread's last exchange failure: which rank was waited for / which rank aborted. */
const char *ovm_hip_shard_last_error(void);

/* Built-in small exchange for the ranks of one node: a POSIX shared-memory segment `name` (a unique
 * name per job, e.g. "/ovm-<port>-<pid>"; every rank passes the same one) with two slots of
 * slot_bytes per rank; a rank stores its contribution and a sequence number, the others spin on it
 * (about a microsecond per exchange). create() returns when every rank has attached, and removes
 * the name. Set ovm_shard_t.allgather_host = ovm_hip_shm_allgather and .ctx = the communicator
 * (or call it from a wrapper if ctx is needed for the device exchange as well). */
typedef struct ovm_shm_comm ovm_shm_comm_t;
int ovm_hip_shm_comm_create(const char *name, uint32_t rank, uint32_t world, uint64_t slot_bytes, ovm_shm_comm_t **out);
/* The same over memory the ranks already share (threads of one process): `base` = zero-filled buffer of
 * ovm_hip_shm_segment_bytes(world, slot_bytes) bytes. */
uint64_t ovm_hip_shm_segment_bytes(uint32_t world, uint64_t slot_bytes);
int ovm_hip_shm_comm_attach(void *base, uint32_t rank, uint32_t world, uint64_t slot_bytes, ovm_shm_comm_t **out);
void ovm_hip_shm_comm_destroy(ovm_shm_comm_t *comm);
int ovm_hip_shm_allgather(void *comm, const void *mine, void *all, uint64_t bytes);
/* ovm_shard_t.abort for the built-in exchange (ctx = the communicator), and whether this rank's handle is poisoned. */
int ovm_hip_shm_comm_abort(void *comm, int code);
int ovm_hip_shm_comm_poisoned(const ovm_shm_comm_t *comm);

/* Built-in BULK exchange for the ranks of one node, no collective library: every rank publishes an IPC handle of
 * the allocation its send buffer lives in, the peers map it (once; the mappings are kept) and pull their part with
 * device-to-device copies on their own stream -- over xGMI between GPUs, or between processes that share a GPU. The
 * handles and the barriers travel through `shm` (which must outlive the communicator). Use: ovm_shard_t.ctx = the
 * ipc communicator, .allgather_host = ovm_hip_ipc_allgather_host (forwards to shm), .allgather_device =
 * ovm_hip_ipc_allgather_device, .alltoall_device = ovm_hip_ipc_alltoall_device, .abort = ovm_hip_ipc_abort. Send buffers must be device
 * allocations of this process (hipMalloc, directly or through a pool). */
typedef struct ovm_ipc_comm ovm_ipc_comm_t;
int ovm_hip_ipc_comm_create(ovm_shm_comm_t *shm, uint32_t rank, uint32_t world, ovm_ipc_comm_t **out);
void ovm_hip_ipc_comm_destroy(ovm_ipc_comm_t *comm);
int ovm_hip_ipc_allgather_host(void *comm, const void *mine, void *all, uint64_t bytes);
int ovm_hip_ipc_allgather_device(void *comm, const void *d_mine, void *d_all, uint64_t bytes, ovm_stream_t stream);
int ovm_hip_ipc_alltoall_device(void *comm, const void *d_send, void *d_recv, uint64_t bytes, ovm_stream_t stream);
int ovm_hip_ipc_abort(void *comm, int code);
/* Mapping a peer's buffer is the one step of this exchange that depends on the driver's IPC path. It is guarded: the ranks
 * map IN TURN (never two processes inside hipIpcOpenMemHandle at once), every open has a deadline (OVM_IPC_OPEN_TIMEOUT_S,
 * default 20 s), and a failed or overdue open -- agreed by all ranks -- DEGRADES the communicator: this and every later
 * bulk exchange is staged through pinned host memory and the shared-memory all-gather (slow, never stuck); the proof in
 * flight completes. ovm_hip_ipc_comm_status reports which carrier a communicator is on and why.
 * One cause is known and avoided outright: on this runtime (ROCm 7.2, dmabuf IPC) hipIpcOpenMemHandle never returns for an
 * allocation whose SIZE has bit 31 set (2-4 GiB, 6-8 GiB, ...: tools/ipc_repro.py). A send buffer inside such an allocation
 * is copied device-to-device into a buffer of the library's own with a safe size, and that one is exported
 * (`bounced_exchanges`; OVM_IPC_NO_BOUNCE=1 turns the rule off for the reproducer). */
typedef struct {
    uint32_t degraded;          /* 1: bulk exchanges are staged through host memory */
    uint32_t mappings;          /* peer allocations currently mapped */
    uint64_t opens;             /* hipIpcOpenMemHandle calls made so far */
    uint64_t exchanges;         /* bulk exchanges that went through the mapped path's set-up */
    uint64_t staged_exchanges;  /* bulk exchanges staged through host memory */
    uint64_t bounced_exchanges; /* exchanges whose send buffer was exported as a device-side copy (see below) */
    char reason[192];           /* why it degraded (empty otherwise) */
} ovm_ipc_status_t;
int ovm_hip_ipc_comm_status(const ovm_ipc_comm_t *comm, ovm_ipc_status_t *out);

/* Host-only self-test of the sharded LogUp-GKR bookkeeping (csrc/gkr_shard.h): `world` simulated ranks
 * (threads over the shared-memory exchange) own the leaf runs [0, ends[0]), [ends[0], ends[1]), ... of a
 * random fraction tree of 2^TR leaves, the rest is padding; every rank's transcript must equal that of the
 * unsharded run with the padding written out, and the windowed eq-heap plans are replayed on the host.
 * Needs no device. 0 = equal; message of a failure: ovm_hip_selftest_error(). */
int ovm_hip_gkr_shard_selftest(uint32_t world, uint32_t TR, const uint64_t *ends, uint64_t seed, int three_points);
const char *ovm_hip_selftest_error(void);
/* Host-only self-test of a communicator: `rounds` small exchanges of round- and rank-dependent sizes (some
 * larger than one shared-memory slot) through allgather_host, and -- if world > 1 and n > 0 -- one bulk
 * exchange of n extension-field elements through allgather_device (and, when alltoall_device is set, the
 * all-to-all + add + all-gather form of the same sum), called with HOST pointers and a null stream, whose
 * blocks are then added up with the field's addition and compared with the expected sum.
 * This is how the CPU tests exercise the callback a host language supplies (tests/test_shard_cpu.py: gloo). */
int ovm_hip_shard_exchange_selftest(const ovm_shard_t *shard, uint64_t n, uint32_t rounds, uint64_t seed);

/* `Coordinator::prove` (prover/mod.rs:104-197) as ONE joint proof computed by shard->world ranks.
 * `traces` lists ALL traces of the proof on every rank (air_idx and height always valid);
 * owner[t] = the rank that holds trace t: only there are d_common_main, public_values and the cached
 * handles read (elsewhere they may be null / zero). Owners must be assigned to the traces in
 * blocks: sorted as the prover sorts them (height descending, air index ascending), owner must be
 * non-decreasing. Every rank calls this with the same key and trace list.
 * Work split: per-trace work (cached-main commitments supplied by the caller, LogUp leaves and the
 * sub-tree over them, round 0, multilinear rounds, its columns of the stacked reduction and of the
 * mu-batch) on the owner; the Merkle trees of the common-main commitment and of the WHIR rounds are
 * built by all ranks together; the WHIR folds and the transcript are replicated on every rank.
 * The proof (identical to ovm_hip_prove's on one device, byte for byte) is written on rank 0 only;
 * other ranks return 0 with *proof_len = 0. `sponge` (optional) is advanced on EVERY rank (the transcript is
 * replicated, the final states are identical). Failures: a rank whose inputs are invalid makes ALL ranks return its
 * code before any work (communicator still usable); a rank that fails later calls shard->abort (if set), the others
 * return OVM_ESHARD_PEER_ABORTED and the communicator is poisoned. On any failure the caller's sponge is untouched. */
int ovm_hip_prove_sharded(const ovm_pk_t *pk, const ovm_shard_t *shard, uint32_t n_traces, const ovm_trace_t *traces,
                          const uint32_t *owner, ovm_sponge_t *sponge, uint8_t *proof_out, uint64_t proof_cap,
                          uint64_t *proof_len, uint64_t *section_offsets5, ovm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENVM_HIP_H */
