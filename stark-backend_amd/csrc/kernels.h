// Internal launch API of the sumcheck-stage kernels (stages.hip). All pointers are device
// pointers unless noted; extension-field buffers are 16-byte aligned arrays of [u32;4].
#pragma once
#include "device_common.h"
#include "device_ctx.h"
#include "program.h"
#include "tuning.h"

#include <utility>
#include <vector>

namespace ovm {

struct DevProgram {
    const Instr *code;
    uint32_t n_instr;
    uint32_t num_slots;
    const EF *weights;
};

// Cached-column program (program.h: stage_cached), device arrays: round 0 and the LogUp-GKR leaves.
struct DevCached {
    const StagedInstr *words = nullptr;  // n_words (a multiple of 8) + one more group of four
    const uint32_t *loads = nullptr;     // records of CachedProgram::MAX_LOADS (column, row offset) pairs
    uint32_t n_words = 0;
    uint32_t num_rows = 0;    // LDS rows of the image (cached columns and live values share them)
    uint32_t rows = 0;        // trace rows per lane the offsets were scaled for
};
// Rows per lane and LDS layout for the cached form; rows == 0: the image does not fit (plain interpreter).
struct CachedPlan {
    uint32_t rows = 0, image_rows = 0, max_loads = 0;
};
// min_rows: the rows the program cannot do without (program.h: cached_liveness)
CachedPlan round0_cached_plan(uint32_t log_lifted, uint32_t n_out, uint32_t min_rows, uint32_t n_instr = 0);
// the leaf program: one row per lane, at most 4 columns brought per block (cached_vm.cuh: the leaf block statement)
CachedPlan leaves_cached_plan(uint32_t min_rows);

// ro.out[k] = sum_b ro.partials[b * K + k], then the flag (for reductions with many outputs)
int sum_partials(const RedOut &ro, uint32_t n_blocks, uint32_t K, hipStream_t s);

int build_selectors(Fp *sels, uint32_t log_lifted, uint32_t log_height, hipStream_t s);
int eq_table(const EF *d_point, uint32_t n, EF *out, hipStream_t s);  // out[y] = prod_i (y_i ? x_i : 1-x_i)
int mobius_eq_table(const EF *d_point, uint32_t n, EF *out, hipStream_t s);

// Coset low-degree extension of the used columns of one trace view.
//   lde[(k * n_cols_total + col) << log_lifted | p] for k < n_cosets, col in used_cols
//   d_shift_pows[k * 2^l_skip + i] = shift_k^i / 2^l_skip
// row0 != 0: the slab of rows [row0, row0 + 2^log_lifted) of the columns (a multiple of 2^l_skip) is extended
// into an lde buffer laid out for 2^log_lifted rows -- round 0 in slabs (zerocheck_save_memory)
// l_skip = 4 (the register kernel): an entry of d_used with bit 31 set also produces the column's rotated view
// d_twins[col] from the same read (d_lag[k * 16 + z] = L_15(shift_k * omega^z), see coset_lde16_kernel); such views are
// then left out of the list. (rotated_lde_list builds it; other l_skip: plain list, the views extended on their own.)
int coset_lde(const ColDesc *d_cols, const uint32_t *d_used, uint32_t n_used, uint32_t n_cols_total,
              uint32_t log_lifted, uint32_t l_skip, uint32_t n_cosets, const Fp *d_shift_pows, Fp *lde,
              hipStream_t s, uint64_t row0 = 0, uint32_t pad_lds_bytes = 0, const int32_t *d_twins = nullptr,
              const Fp *d_lag = nullptr);
// (pad_lds_bytes: LDS every workgroup claims without using it -- caps the workgroups per CU of a launch that runs beside
// latency-critical kernels of another stream)

// Round 0: out[((k * n_out) + o) * N + z] = sum_x eq_xi[x] * program_o(lde_k row (x<<l)+z) (+ add_const for o==1)
// Round 0 of many traces in ONE launch (staged programs of equal rows-per-lane and output count): the
// small traces of a many-AIR proof are otherwise dozens of launches of a few workgroups each. Job j
// owns workgroups [wg0, wg0 + gx); its last workgroup sums the job's records into job.out (K =
// n_cosets * n_out << l_skip mailbox slots), the last job raises ro.flag. d_counters: n_jobs + 1
// words, zero on entry and on exit.
struct R0Job {
    DevCached prog;
    const Fp *lde;
    const EF *eq_xi;
    EF *partials;
    EF *out;
    EF add_const;
    uint32_t n_cols_total, log_lifted, n_cosets, gx;
    uint32_t wg0;
    // a long program on a short trace is split into n_groups sub-programs (disjoint sets of accumulation
    // roots, the sums add): workgroup bx of the job runs group bx / gx_group; progs = device array of their
    // staged forms (n_groups == 1: `prog` alone). lds_rows = the largest num_rows among them.
    uint32_t n_groups, gx_group, lds_rows;
    const DevCached *progs;
};
// workgroups round0_eval gives a cached-form program on a trace of 2^log_lifted rows
uint32_t round0_cached_grid(const DevCached &staged, uint32_t log_lifted);
int round0_eval_batched(const R0Job *d_jobs, const R0Job *h_jobs, uint32_t n_jobs, uint32_t rows, uint32_t n_out,
                        uint32_t l_skip, uint32_t *d_counters, const RedOut &ro, hipStream_t s);

// `staged` (optional, words != nullptr): the same program in cached-column form; used when its LDS image fits.
int round0_eval(const DevProgram &prog, const DevCached &staged, const Fp *lde, uint32_t n_cols_total, uint32_t log_lifted, uint32_t l_skip,
                uint32_t n_cosets, uint32_t n_out, const EF *d_eq_xi, EF add_const, const RedOut &ro, hipStream_t s);

// out[col * (lifted >> l) + x] = sum_z w[z] * col(( x << l) + z)
// d_twins (optional, n_cols entries): see fold_ple_kernel -- a column and its rotated view computed from one read
int fold_ple(const ColDesc *d_cols, const int32_t *d_twins, uint32_t n_cols, uint32_t log_lifted, uint32_t l_skip,
             const EF *d_weights, EF *out, hipStream_t s);

// MLE round on folded EF matrices (n_cols x 2^log_h, column-major), node-grouped program:
//   out[2*(X-1)+o] = sum_y eq[y] * program_o(row interpolated at X from rows 2y, 2y+1);
//   grid = (y blocks, X = 1..degree, group).
// d_words / d_woff / lds_rows / d_weights_pre: the same groups in accumulator form (program.h: MleWords, MleWeight) for the
// LDS-slot executor, taken when lds_rows x degree fits (mle_lds_eligible); d_words may be null (slots in private memory).
int mle_round_eval_grouped(const Instr *d_code, const uint32_t *d_off, uint32_t n_groups, uint32_t max_slots, const MleWord *d_words,
                           const uint32_t *d_woff, uint32_t lds_rows, const EF *d_weights, const MleWeight *d_weights_pre,
                           const EF *mats, uint32_t log_h, uint32_t degree, uint32_t n_out, const EF *d_eq, EF add_const,
                           const RedOut &ro, hipStream_t s);

// The same for many (trace, program) jobs in ONE launch: the late rounds of a many-AIR proof are
// dozens of launches of a handful of workgroups each, serialised by the stream; batched they run
// side by side. blockIdx.z enumerates (job, group); job j's last workgroup sums its records into
// job.out (a mailbox slot), the last job to finish raises ro.flag. d_counters: n_jobs + 1 words,
// zero on entry, zero again on exit.
struct MleJob {
    const Instr *code;
    const uint32_t *off;
    const EF *weights;
    const EF *mats;
    const EF *eq;
    EF *partials;
    EF *out;
    EF add_const;
    uint32_t log_h, n_out, n_groups, gx;
    uint32_t z0, max_slots;   // max_slots: what an executor that ignores the forwarding hints needs (program.h: compact_slots)
    uint32_t partials_cap, lds_slots;  // capacity of `partials` in EF (checked by the launcher); LDS rows of the words
    // the accumulator form for the LDS-slot kernel (program.h: MleWords; null for jobs of the other kernel)
    const MleWord *words;
    const uint32_t *woff;
    const MleWeight *weights_pre;
};
int mle_round_eval_batched(const MleJob *d_jobs, const MleJob *h_jobs, uint32_t n_jobs, uint32_t degree,
                           uint32_t *d_counters, const RedOut &ro, hipStream_t s);
// The same for jobs that qualify for the LDS-slot interpreter (mle_lds_eligible) and have at least a workgroup of pairs:
// lanes enumerate pairs, all points in the lane; job.gx from mle_round_gx(log_h, degree, n_groups, true).
int mle_round_eval_batched_lds(const MleJob *d_jobs, const MleJob *h_jobs, uint32_t n_jobs, uint32_t degree,
                               uint32_t *d_counters, const RedOut &ro, hipStream_t s);
bool mle_lds_eligible(uint32_t lds_slots, uint32_t degree);
// workgroups per (job, group) of the batched kernel (lanes enumerate (pair, point))
uint32_t mle_round_batched_gx(uint32_t log_h, uint32_t degree, uint32_t n_groups);
// workgroups along y that mle_round_eval_grouped uses for 2^(log_h-1) pairs and this many groups
uint32_t mle_round_gx(uint32_t log_h, uint32_t degree, uint32_t n_groups);

// out[c * h/2 + y] = in[c*h + 2y] + r * (in[c*h + 2y + 1] - in[c*h + 2y])
int fold_mle(const EF *in, EF *out, uint32_t n_cols, uint32_t log_h, EF r, hipStream_t s);

// fold_mle for many small matrices in one launch (blockIdx.y enumerates (job, column))
struct FoldJob {
    const EF *in;
    EF *out;
    uint32_t n_cols, log_h, y0, pad;
};
// d_jobs == nullptr: the jobs (at most FoldJobsArg::MAX) are passed to the kernel as an argument -- no upload
struct FoldJobsArg {
    static constexpr uint32_t MAX = 16;
    FoldJob j[MAX];
};
int fold_mle_batched(const FoldJob *d_jobs, const FoldJob *h_jobs, uint32_t n_jobs, EF r, hipStream_t s);

// merkle.hip: one rank's share of a Merkle tree that dist_n ranks build together (see there)
int merkle_tree_share(const uint32_t *d_matrix, uint64_t width, uint32_t log_rows_share, uint32_t log_rows_per_query,
                      uint32_t *d_scratch, uint32_t *d_digest_layers, hipStream_t stream, bool ext, uint32_t dist_n,
                      uint32_t dist_r);

// ---- LogUp GKR ----  (layer format: record x = fractions 2x, 2x+1 = (p0,q0),(p1,q1))
// Heap of partial eq tables of one layer: depth k holds eq(point[n-k .. n), y) for the LAST k variables (node 1 = 1):
// the weights of sub-round n-1-k of a layer sumcheck whose eq factor is split off (stages.hip).
// Windowed form for a rank that owns an index window of the layer's table (sharded GKR, gkr_shard.h):
// need[d] = [lo, hi) = the entries of depth d the sub-round sums read. Depths <= top are full tables at
// heap index 2^d + y; a deeper depth d stores entries [lo[d], lo[d] + cnt[d]) at heap index off[d] + (y - lo[d]).
// With need[d] = [0, 2^d) this is exactly the heap eq_heap builds. total = heap elements needed.
struct EqHeapPlan {
    uint32_t n = 0, top = 0;
    bool empty = true;
    uint64_t total = 0;
    std::vector<uint64_t> off, lo, cnt;
    struct Launch {
        uint32_t k, levels;
        uint64_t src_first, count;
    };
    std::vector<Launch> launches;
    // heap index of entry y of depth d
    uint64_t at(uint32_t d, uint64_t y) const { return off[d] + (y - lo[d]); }
};
EqHeapPlan eq_heap_plan(uint32_t n, const std::vector<std::pair<uint64_t, uint64_t>> &need);
// `point`: HOST array of plan.n entries (it travels to the kernels as an argument, plan.n <= EqPoint::MAX)
struct EqPoint {
    static constexpr uint32_t MAX = 32;
    EF x[MAX];
};
int eq_heap_run(const EF *point, const EqHeapPlan &plan, EFPlanes heap, hipStream_t s);
// fractions at arbitrary plane positions, for the host (n <= 32)
int gkr_read_list(FracPlanes t, const uint64_t *d_pos, const uint32_t *d_pbase, uint32_t n, const RedOut &ro, hipStream_t s);
// table records for the host: n0 consecutive ones from plane offset at0, then n1 from at1 (n0 + n1 <= 64), read
// as they are (folds = 0: two fractions per record), folded by r1 (1: four fractions per record) or folded by r1 and then r2
// (2: eight fractions per record)
int gkr_edge_records(FracPlanes src, uint64_t at0, uint64_t at1, uint32_t n0, uint32_t n1, bool p_base, int folds, EF r1, EF r2,
                     const RedOut &ro, hipStream_t s);
// The same records as a rider of the sub-round's sums kernel (gkr_round / gkr_fold_round): its workgroup 0 writes them to
// `out` (mailbox slots, 4 EF per record) before its own pairs; the kernel's flag covers them. n0 + n1 = 0: none.
struct GkrEdges {
    FracPlanes src;
    uint64_t at0 = 0, at1 = 0;
    uint32_t n0 = 0, n1 = 0;
    int p_base = 0, folds = 0;
    EF r1 = EF::zero(), r2 = EF::zero();
    EF *out = nullptr;
};
// p_base: leaf numerators are base-field, only limb plane 0 of P is written / read (stages.hip)
// The leaf kernels of many traces in one launch (staged programs, one row per lane): job j owns
// workgroups [wg0, wg0 + gx).
struct LeavesJob {
    DevCached prog;
    const EF *leaf_const;
    const ColDesc *cols;
    const uint64_t *leaf_base;
    const uint32_t *leaf_log_len;
    Fp norm;
    uint32_t log_height, gx, wg0;
};
int gkr_eval_leaves_batched(const LeavesJob *d_jobs, const LeavesJob *h_jobs, uint32_t n_jobs, FracPlanes leaves,
                            hipStream_t s);
// `staged` (optional): the DEN/LEAF program in staged form, laid out for one row per lane
int gkr_eval_leaves(const DevProgram &prog, const DevCached &staged, const EF *d_leaf_const, const ColDesc *d_cols, uint32_t log_height,
                    const uint64_t *d_leaf_base, const uint32_t *d_leaf_log_len, Fp norm, FracPlanes leaves,
                    hipStream_t s);
int gkr_build_layer(FracPlanes child, FracPlanes parent, uint64_t n_parent, bool child_p_base, hipStream_t s);
// three levels in one pass; pair_*: grandparents without a sibling in the range (with their two parents), lone_*: parents without one
int gkr_build_three_layers(FracPlanes child, FracPlanes parent, FracPlanes grand, FracPlanes great, uint64_t n_great,
                           const FracPlanes *pair_child, const FracPlanes *pair_parent, const FracPlanes *pair_grand, uint32_t n_pair,
                           const FracPlanes *lone_child, const FracPlanes *lone_parent, uint32_t n_lone, bool child_p_base,
                           hipStream_t s);
// out[X-1] = sum_y eq[y] * (p0 q1 + p1 q0 + lambda q0 q1)(X, y), X = 1..n_points (2 or 3; eq: n_pairs weights)
int gkr_round(FracPlanes rec, bool p_base, EqSplit eq, uint64_t n_pairs, EF lambda, uint32_t n_points,
              const RedOut &ro, hipStream_t s, const GkrEdges &ed = GkrEdges());
// fold the previous round with r and return the NEXT round's sums over the n_pairs folded pairs
int gkr_fold_round(FracPlanes in, bool in_p_base, EqSplit eq, FracPlanes out, uint64_t n_pairs, EF r, EF lambda,
                   uint32_t n_points, const RedOut &ro, hipStream_t s, const GkrEdges &ed = GkrEdges());

// ---- stacked reduction ----
// R[v * N + z] for v in {eq, rot1, rot2} over one window of `n_slices` contiguous slices of
// 2^log_len rows each starting at `base`.
int stacked_round0(const Fp *base, uint32_t n_slices, uint32_t log_len, uint32_t l_skip, const EF *d_lam_eq,
                   const EF *d_lam_rot, const EF *d_eq_cube, const RedOut &ro, hipStream_t s);
// Q[v][x] = sum_i coef_v[i] * q[(col_i * h + row_i + x)] for v in {eq, rot}
// d_coef_*_pre: 8 words per slice = {c0, c1, c2, c3, 0, 11 c1, 11 c2, 11 c3} (Montgomery words; host: coef_pre)
// chunks > 1: the slices are split over `chunks` workgroup rows whose partial sums go through d_scratch
// (chunks * 2 << log_len EF) -- for short windows of thousands of slices (stacked_round0_chunks)
uint32_t stacked_round0_chunks(uint32_t n_slices, uint32_t log_len);
int stacked_combine(const EF *q, const uint64_t *d_slice_off, uint32_t n_slices, uint32_t log_len,
                    const uint32_t *d_coef_eq_pre, const uint32_t *d_coef_rot_pre, EF *out_eq, EF *out_rot, EF *d_scratch,
                    uint32_t chunks, hipStream_t s);
// s[X-1] = sum_y Qeq(X,y) eq(X,y) + Qrot(X,y) krot(X,y), X = 1,2 ; vec = 4 columns of 2^log_h
int stacked_round(const EF *vec4, uint32_t log_h, const RedOut &ro, hipStream_t s);
// The round's two sums over the slices that have no variables left (round > n_lift), D_i kept in d_tail (see stages.hip).
struct TailSlice {
    uint64_t row;      // row_idx of the slice in its stacked column
    uint32_t gcol;     // its column among this rank's stacked columns
    uint32_t n_lift;   // its own rounds; ~0: not this rank's
};
int stacked_tail(const EF *q, uint32_t q_log_h, const TailSlice *d_slices, EF *d_tail, uint32_t n, uint32_t l_skip, uint32_t round,
                 EF u_prev, const RedOut &ro, hipStream_t s);

// ---- WHIR ----
int whir_mu_batch(const Fp *msg, uint32_t width, uint32_t log_h, const EF *d_mu_pows, uint32_t mu_offset,
                  EF *f, int accumulate, hipStream_t s);
// in-place subset-zeta (dir=+1: coeffs->evals) or Moebius (dir=-1: evals->coeffs) over all bits
int zeta_transform(EF *v, uint32_t log_n, int dir, hipStream_t s);
int whir_round(const EF *f, const EF *w, uint64_t n_pairs, const RedOut &ro, hipStream_t s);
// fold the previous round with alpha and return the NEXT round's sums over the n_pairs folded pairs
int whir_fold_round(const EF *f_in, const EF *w_in, EF *f_out, EF *w_out, uint64_t n_pairs, EF alpha, const RedOut &ro,
                    hipStream_t s);
int whir_fold(const EF *f_in, const EF *w_in, EF *f_out, EF *w_out, uint64_t n_out, EF alpha, hipStream_t s);
// out = sum_i coeffs[i] * z^i
int horner_eval(const EF *coeffs, uint64_t n, EF z, const RedOut &ro, hipStream_t s);
// w[x] += sum_q gamma_q * eq(x, pow(z_q)) using split tables lo/hi (n_q x 2^half each)
int whir_w_accumulate(EF *w, uint32_t log_n, const EF *d_tab_lo, const EF *d_tab_hi, uint32_t log_lo,
                      uint32_t n_q, hipStream_t s);
// out[i] = sum_b in[b * n + i]
int ef_sum_blocks(const EF *in, uint32_t n_blocks, uint64_t n, EF *out, hipStream_t s);
int gather_ef(const EF *src, const uint64_t *d_idx, uint64_t n, EF *out, hipStream_t s);

}  // namespace ovm
