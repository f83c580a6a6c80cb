// The cached-column round-0 interpreter (program.h: stage_cached), device code for gfx950.
//
// Replaces (reference): the rule interpreter of crates/cuda-backend/cuda/src/logup_zerocheck/zerocheck_round0.cu:30-137
// (acc_constraints: decode a rule, fetch operands, operate, buffer, accumulate) for the coset evaluations of round 0.
//
// One lane evaluates R trace rows. State of a lane: the RESULT REGISTER (R words: the value of the word before), the
// accumulators (96-bit sums of weight * value, eval.cuh: Acc96), the LANDING registers (up to 8 columns x R cells
// requested from the coset LDE one block ago) and its words of the workgroup's LDS image: row-major, word
// (row * R + r) * 256 + lane, rows = cache of columns + the program's slots.
//
// The program is a stream of 16-byte words in scalar registers (fetched four at a time, one group ahead; a whole number
// of pairs of groups):
//   arithmetic  {StagedOp | flags, dst, a, b}: first operand = the result register (S_FWD_A) or LDS row a, loaded INTO the
//               result register; second = constant b (S_B_CONST), the result register (S_FWD_B), or LDS row b; the result
//               replaces the result register and is stored to row dst unless S_NOSTORE
//   C_ACC       accumulator (bit 31 of .y) += weight (the four words, bit 31 cleared) * result register
//   C_BLOCK     (first word of a group only) the landing registers go to their rows, the next block's columns are requested
//
// Why assembly. Written as C++ the compiler turns every uniform `first operand = forwarded ? result : LDS` into per-cell
// branches with register copies, copies all 12 R accumulator registers in and out of the accumulation path, and waits for
// the landing registers' loads at the first arithmetic word after they were requested (the whole point is that they stay
// in flight for a block). Each piece below is ONE asm statement that updates its registers in place and skips itself
// with an internal branch, so the C++ around it is straight-line code: no phi, no copy, and every s_waitcnt is placed by
// hand. Hazards checked against the gfx940 list: SMEM results are waited for before use, SALU-written SGPRs feed VMEM
// addresses (no wait states needed, unlike VALU-written ones), nothing writes M0 / EXEC / VCC across statements.
#pragma once
#include "eval.cuh"
#include "kernels.h"

namespace ovm {

#if defined(__HIP_DEVICE_COMPILE__)
#define OVM_ROWS_1(M) M(0)
#define OVM_ROWS_2(M) M(0) M(1)
#define OVM_ROWS_4(M) M(0) M(1) M(2) M(3)
#define OVM_OFF_0 "0"
#define OVM_OFF_1 "1024"
#define OVM_OFF_2 "2048"
#define OVM_OFF_3 "3072"
static_assert(RED_THREADS * 4 == 1024, "the asm addresses row r of an operand at byte offset r * 1024");

// second operand of row r: an LDS value, the constant (a scalar register), the result register itself
#define OVM_SRC_B(r) "%[b" #r "]"
#define OVM_SRC_C(r) "%[wb]"
#define OVM_SRC_P(r) "%[p" #r "]"
// p = p + s, p - s, s - p, p * s (Montgomery) mod P; kP = P, kM = P^-1 mod 2^32; t0, t1 shared by the rows
#define OVM_ALU_ADD(r, S) \
    "v_add_u32 %[t0], " S(r) ", %[p" #r "]\n\tv_subrev_u32 %[t1], %[kP], %[t0]\n\tv_min_u32 %[p" #r "], %[t0], %[t1]\n\t"
#define OVM_ALU_SUB(r, S) \
    "v_subrev_u32 %[t0], " S(r) ", %[p" #r "]\n\tv_add_u32 %[t1], %[kP], %[t0]\n\tv_min_u32 %[p" #r "], %[t0], %[t1]\n\t"
#define OVM_ALU_RSUB(r, S) \
    "v_sub_u32 %[t0], " S(r) ", %[p" #r "]\n\tv_add_u32 %[t1], %[kP], %[t0]\n\tv_min_u32 %[p" #r "], %[t0], %[t1]\n\t"
#define OVM_ALU_MUL(r, S)                                                                                          \
    "v_mul_lo_u32 %[t0], %[p" #r "], " S(r) "\n\tv_mul_hi_u32 %[t1], %[p" #r "], " S(r) "\n\t"                      \
    "v_mul_lo_u32 %[t0], %[t0], %[kM]\n\tv_mul_hi_u32 %[t0], %[t0], %[kP]\n\tv_sub_u32 %[t1], %[t1], %[t0]\n\t"     \
    "v_add_u32 %[t0], %[kP], %[t1]\n\tv_min_u32 %[p" #r "], %[t1], %[t0]\n\t"
#define OVM_ADD_B(r) OVM_ALU_ADD(r, OVM_SRC_B)
#define OVM_ADD_C(r) OVM_ALU_ADD(r, OVM_SRC_C)
#define OVM_ADD_P(r) OVM_ALU_ADD(r, OVM_SRC_P)
#define OVM_SUB_B(r) OVM_ALU_SUB(r, OVM_SRC_B)
#define OVM_SUB_C(r) OVM_ALU_SUB(r, OVM_SRC_C)
#define OVM_SUB_P(r) OVM_ALU_SUB(r, OVM_SRC_P)
#define OVM_RSUB_B(r) OVM_ALU_RSUB(r, OVM_SRC_B)
#define OVM_RSUB_C(r) OVM_ALU_RSUB(r, OVM_SRC_C)
#define OVM_RSUB_P(r) OVM_ALU_RSUB(r, OVM_SRC_P)
#define OVM_MUL_B(r) OVM_ALU_MUL(r, OVM_SRC_B)
#define OVM_MUL_C(r) OVM_ALU_MUL(r, OVM_SRC_C)
#define OVM_MUL_P(r) OVM_ALU_MUL(r, OVM_SRC_P)
#define OVM_MOVC(r) "v_mov_b32 %[p" #r "], %[wb]\n\t"
#define OVM_RD_P(r) "ds_read_b32 %[p" #r "], %[ad] offset:" OVM_OFF_##r "\n\t"
#define OVM_RD_B(r) "ds_read_b32 %[b" #r "], %[ad] offset:" OVM_OFF_##r "\n\t"
#define OVM_WR_P(r) "ds_write_b32 %[ad], %[p" #r "] offset:" OVM_OFF_##r "\n\t"
// accumulator A (0 / 1) of row r += weight * result register: four 96-bit multiply-adds
#define OVM_MAC(A, r, k, W)                                                                            \
    "v_mad_u64_u32 %[l" #A #r #k "], vcc, %[p" #r "], " W ", %[l" #A #r #k "]\n\t"                      \
    "v_addc_co_u32_e32 %[h" #A #r #k "], vcc, 0, %[h" #A #r #k "], vcc\n\t"
#define OVM_MACS(A, r) OVM_MAC(A, r, 0, "%[ts0]") OVM_MAC(A, r, 1, "%[ts1]") OVM_MAC(A, r, 2, "%[wa]") OVM_MAC(A, r, 3, "%[wb]")
#define OVM_MACS0(r) OVM_MACS(0, r)
#define OVM_MACS1(r) OVM_MACS(1, r)

// One word (not a block word): an accumulation or an arithmetic word. ROWS = OVM_ROWS_<R>; ACC1 = the code of accumulator
// 1 (NOUT = 2) or "". Branch targets: 20 arithmetic, 30 / 40 second operand constant / result register, 5x the operations
// on an LDS second operand, 6x on the constant, 7x on the result register, 80 store, 90 end.
// S_MUL = 1, S_ADD = 2, S_SUB = 4, S_RSUB = 8, S_MOV = 32, S_MOVC = 64 (bits 0, 1, 2, 3, 5, 6); S_B_CONST bit 8, S_FWD_A 9,
// S_FWD_B 10, S_NOSTORE 11 (program.h).
#define OVM_WORD_ASM(ROWS, ACC1)                                                                                     \
    "s_cmp_lt_u32 %[wx], 0x40000000\n\ts_cbranch_scc1 20f\n\t"                                                       \
    "s_bitcmp1_b32 %[wx], 31\n\ts_cbranch_scc0 90f\n\t"                                                              \
    "s_and_b32 %[ts0], %[wx], 0x7fffffff\n\ts_and_b32 %[ts1], %[wy], 0x7fffffff\n\t"                                 \
    ACC1                                                                                                             \
    ROWS(OVM_MACS0)                                                                                                  \
    "s_branch 90f\n"                                                                                                 \
    "20:\n\t"                                                                                                        \
    "s_bitcmp1_b32 %[wx], 9\n\ts_cbranch_scc1 21f\n\t"                                                               \
    "v_lshl_add_u32 %[ad], %[wa], 2, %[mine]\n\t" ROWS(OVM_RD_P)                                                     \
    "21:\n\t"                                                                                                        \
    "s_bitcmp1_b32 %[wx], 8\n\ts_cbranch_scc1 30f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 10\n\ts_cbranch_scc1 40f\n\t"                                                              \
    "v_lshl_add_u32 %[ad], %[wb], 2, %[mine]\n\t" ROWS(OVM_RD_B)                                                     \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                       \
    "s_bitcmp1_b32 %[wx], 1\n\ts_cbranch_scc1 51f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 2\n\ts_cbranch_scc1 52f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 0\n\ts_cbranch_scc1 53f\n\t"                                                               \
    ROWS(OVM_RSUB_B) "s_branch 80f\n"                                                                                \
    "51:\n\t" ROWS(OVM_ADD_B) "s_branch 80f\n"                                                                       \
    "52:\n\t" ROWS(OVM_SUB_B) "s_branch 80f\n"                                                                       \
    "53:\n\t" ROWS(OVM_MUL_B) "s_branch 80f\n"                                                                       \
    "30:\n\t"                                                                                                        \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                       \
    "s_bitcmp1_b32 %[wx], 1\n\ts_cbranch_scc1 61f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 2\n\ts_cbranch_scc1 62f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 0\n\ts_cbranch_scc1 63f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 5\n\ts_cbranch_scc1 80f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 3\n\ts_cbranch_scc1 64f\n\t"                                                               \
    ROWS(OVM_MOVC) "s_branch 80f\n"                                                                                  \
    "61:\n\t" ROWS(OVM_ADD_C) "s_branch 80f\n"                                                                       \
    "62:\n\t" ROWS(OVM_SUB_C) "s_branch 80f\n"                                                                       \
    "63:\n\t" ROWS(OVM_MUL_C) "s_branch 80f\n"                                                                       \
    "64:\n\t" ROWS(OVM_RSUB_C) "s_branch 80f\n"                                                                      \
    "40:\n\t"                                                                                                        \
    "s_bitcmp1_b32 %[wx], 1\n\ts_cbranch_scc1 71f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 0\n\ts_cbranch_scc1 73f\n\t"                                                               \
    "s_bitcmp1_b32 %[wx], 2\n\ts_cbranch_scc1 72f\n\t"                                                               \
    ROWS(OVM_RSUB_P) "s_branch 80f\n"                                                                                \
    "71:\n\t" ROWS(OVM_ADD_P) "s_branch 80f\n"                                                                       \
    "72:\n\t" ROWS(OVM_SUB_P) "s_branch 80f\n"                                                                       \
    "73:\n\t" ROWS(OVM_MUL_P)                                                                                        \
    "80:\n\t"                                                                                                        \
    "s_bitcmp1_b32 %[wx], 11\n\ts_cbranch_scc1 90f\n\t"                                                              \
    "v_lshl_add_u32 %[ad], %[wy], 2, %[mine]\n\t" ROWS(OVM_WR_P)                                                     \
    "90:"
#define OVM_ACC1_ASM(ROWS) "s_bitcmp1_b32 %[wy], 31\n\ts_cbranch_scc0 10f\n\t" ROWS(OVM_MACS1) "s_branch 90f\n10:\n\t"

// Block word. The record (16 words: 8 row offsets, then 8 flat columns) is read into s[84:99] -- fixed registers, free
// inside this one statement (clobbered) -- because a 16-register operand cannot be indexed from the asm text.
// s[80:81]: column address; s82 / s83: loads to land / to request.
#define OVM_LAND_1(j, SR) "v_lshl_add_u32 %[ad], " SR ", 2, %[mine]\n\tds_write_b32 %[ad], %[g" #j "0]\n\t"
#define OVM_LAND_2(j, SR) OVM_LAND_1(j, SR) "ds_write_b32 %[ad], %[g" #j "1] offset:1024\n\t"
#define OVM_LAND_4(j, SR) OVM_LAND_2(j, SR) "ds_write_b32 %[ad], %[g" #j "2] offset:2048\n\tds_write_b32 %[ad], %[g" #j "3] offset:3072\n\t"
#define OVM_COLADDR(SC)                                                                        \
    "s_mul_hi_u32 s81, " SC ", %[pow]\n\ts_mul_i32 s80, " SC ", %[pow]\n\t"                     \
    "s_add_u32 s80, s80, %[ldelo]\n\ts_addc_u32 s81, s81, %[ldehi]\n\t"
#define OVM_REQ_1(j, SC) OVM_COLADDR(SC) "global_load_dword %[g" #j "0], %[o0], s[80:81]\n\t"
#define OVM_REQ_2(j, SC) OVM_REQ_1(j, SC) "global_load_dword %[g" #j "1], %[o1], s[80:81]\n\t"
#define OVM_REQ_4(j, SC) OVM_REQ_2(j, SC) "global_load_dword %[g" #j "2], %[o2], s[80:81]\n\tglobal_load_dword %[g" #j "3], %[o3], s[80:81]\n\t"
// The leaf kernel reads trace columns, not a coset LDE: the record's second half is four (pointer lo, pointer hi | rotated
// << 31) pairs -- s[92:93] .. s[98:99] -- and the lane has two offsets, its row (o0) and the row after it (o1).
#define OVM_REQ_LEAF(j, LO, HI)                                                                        \
    "s_bitcmp1_b32 s" HI ", 31\n\ts_bitset0_b32 s" HI ", 31\n\ts_cbranch_scc1 1f\n\t"                    \
    "global_load_dword %[g" #j "0], %[o0], s[" LO ":" HI "]\n\ts_branch 2f\n"                          \
    "1:\n\tglobal_load_dword %[g" #j "0], %[o1], s[" LO ":" HI "]\n"                                   \
    "2:\n\t"
#define OVM_NEXT(S, j1, L) "s_cmp_lt_u32 " S ", " #j1 "\n\ts_cbranch_scc1 " L "\n\t"
#define OVM_BLOCK_ASM(LAND, REQ)                                                                                     \
    "s_lshr_b32 s82, %[wx], 30\n\ts_cmp_lg_u32 s82, 1\n\ts_cbranch_scc1 90f\n\t"                                      \
    "s_lshl_b32 s80, %[wy], 6\n\ts_load_dwordx16 s[84:99], %[recs], s80\n\t"                                          \
    "s_and_b32 s82, %[wx], 0xff\n\ts_bfe_u32 s83, %[wx], 0x80008\n\t"                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                       \
    "s_cmp_eq_u32 s82, 0\n\ts_cbranch_scc1 50f\n\t"                                                                   \
    "s_waitcnt vmcnt(0)\n\t"                                                                                         \
    LAND(0, "s84") OVM_NEXT("s82", 2, "50f") LAND(1, "s85") OVM_NEXT("s82", 3, "50f") LAND(2, "s86") OVM_NEXT("s82", 4, "50f")  \
    LAND(3, "s87") OVM_NEXT("s82", 5, "50f") LAND(4, "s88") OVM_NEXT("s82", 6, "50f") LAND(5, "s89") OVM_NEXT("s82", 7, "50f")  \
    LAND(6, "s90") OVM_NEXT("s82", 8, "50f") LAND(7, "s91")                                                          \
    "50:\n\t"                                                                                                        \
    "s_cmp_eq_u32 s83, 0\n\ts_cbranch_scc1 90f\n\t"                                                                   \
    REQ(0, "s92") OVM_NEXT("s83", 2, "90f") REQ(1, "s93") OVM_NEXT("s83", 3, "90f") REQ(2, "s94") OVM_NEXT("s83", 4, "90f")     \
    REQ(3, "s95") OVM_NEXT("s83", 5, "90f") REQ(4, "s96") OVM_NEXT("s83", 6, "90f") REQ(5, "s97") OVM_NEXT("s83", 7, "90f")     \
    REQ(6, "s98") OVM_NEXT("s83", 8, "90f") REQ(7, "s99")                                                            \
    "90:"
#define OVM_BLOCK_LEAF_ASM(LAND)                                                                                      \
    "s_lshr_b32 s82, %[wx], 29\n\ts_cmp_lg_u32 s82, 2\n\ts_cbranch_scc1 90f\n\t"                                      \
    "s_lshl_b32 s80, %[wy], 6\n\ts_load_dwordx16 s[84:99], %[recs], s80\n\t"                                          \
    "s_and_b32 s82, %[wx], 0xff\n\ts_bfe_u32 s83, %[wx], 0x80008\n\t"                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                       \
    "s_cmp_eq_u32 s82, 0\n\ts_cbranch_scc1 50f\n\t"                                                                   \
    "s_waitcnt vmcnt(0)\n\t"                                                                                         \
    LAND(0, "s84") OVM_NEXT("s82", 2, "50f") LAND(1, "s85") OVM_NEXT("s82", 3, "50f") LAND(2, "s86") OVM_NEXT("s82", 4, "50f")  \
    LAND(3, "s87")                                                                                                   \
    "50:\n\t"                                                                                                        \
    "s_cmp_eq_u32 s83, 0\n\ts_cbranch_scc1 90f\n\t"                                                                   \
    OVM_REQ_LEAF(0, "92", "93") OVM_NEXT("s83", 2, "90f") OVM_REQ_LEAF(1, "94", "95") OVM_NEXT("s83", 3, "90f")         \
    OVM_REQ_LEAF(2, "96", "97") OVM_NEXT("s83", 4, "90f") OVM_REQ_LEAF(3, "98", "99")                                 \
    "90:"
#define OVM_BLOCK_CLOBBERS                                                                                               \
    "memory", "scc", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", \
        "s95", "s96", "s97", "s98", "s99"
#endif  // __HIP_DEVICE_COMPILE__

// The interpreter proper. LEAVES = false: round 0 (columns of a coset LDE: base + column * pow + o[r]); LEAVES = true: the
// LogUp-GKR leaf program (R = 1; trace columns by pointer, o[0] = this row, o[1] = the row after it; `leaf(k)` is called where
// the program closes interaction k, with the result register = its numerator and accumulator 0 = its denominator sum).
template <int R, int NOUT, bool LEAVES>
struct CachedCore {
    static_assert(R == 1 || R == 2 || R == 4, "one asm form per R");
    static_assert(NOUT == 1 || (NOUT == 2 && R <= 2), "two accumulator sets for at most two rows per lane");
    static_assert(!LEAVES || (R == 1 && NOUT == 1), "the leaf program runs one row per lane");
    static_assert(CachedProgram::MAX_LOADS == 8, "the block statement lands / requests 8 columns");
    template <class Leaf>
    static __device__ __forceinline__ void run(const DevCached &prog, uint64_t base_addr, uint32_t pow_in, const uint32_t (&o)[4],
                                               Acc96 (&acc0)[R], Acc96 (&acc1)[R], Leaf &&leaf) {
#if defined(__HIP_DEVICE_COMPILE__)
        extern __shared__ uint32_t lds_image[];
        // LDS byte address of this lane's word of row 0 (the asm adds operand offsets and r * 1024)
        const uint32_t mine = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds_image + threadIdx.x * 4;
        // Everything the statements take in scalar registers is made wave-uniform HERE: an "s" operand that the compiler
        // holds in a vector register (a descriptor read from a job list with vector loads) is passed on as a VGPR name.
        auto uni = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
        auto uni64 = [&](uint64_t x) { return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | uni((uint32_t)x); };
        const OVM_CONST_AS uint4 *wp = (const OVM_CONST_AS uint4 *)uni64((uint64_t)(uintptr_t)prog.words);
        const OVM_CONST_AS uint32_t *recs = (const OVM_CONST_AS uint32_t *)uni64((uint64_t)(uintptr_t)prog.loads);
        const uint32_t n_words = uni(prog.n_words);
        const uint64_t lde_addr = uni64(base_addr);
        const uint32_t lde_lo = (uint32_t)lde_addr, lde_hi = (uint32_t)(lde_addr >> 32);
        const uint32_t pow = uni(pow_in);
        const uint32_t kP = bb::P, kM = bb::MONTY_M;
        uint32_t g[8][R];  // landing registers
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int r = 0; r < R; r++) g[j][r] = 0;
        uint32_t pr[R];  // result register
#pragma unroll
        for (int r = 0; r < R; r++) pr[r] = 0;
        uint32_t ad, t0, t1, ts0, ts1, b[R];

#define OVM_ACC_OPS(A, r)                                                                                                    \
    [l##A##r##0] "+v"(acc##A[r].lo[0]), [l##A##r##1] "+v"(acc##A[r].lo[1]), [l##A##r##2] "+v"(acc##A[r].lo[2]),                    \
        [l##A##r##3] "+v"(acc##A[r].lo[3]), [h##A##r##0] "+v"(acc##A[r].hi[0]), [h##A##r##1] "+v"(acc##A[r].hi[1]),                \
        [h##A##r##2] "+v"(acc##A[r].hi[2]), [h##A##r##3] "+v"(acc##A[r].hi[3])
#define OVM_WORD_IN(W) [wx] "s"(W.x), [wy] "s"(W.y), [wa] "s"(W.z), [wb] "s"(W.w), [mine] "v"(mine), [kP] "s"(kP), [kM] "s"(kM)
#define OVM_WORD_TMP [ad] "=&v"(ad), [t0] "=&v"(t0), [t1] "=&v"(t1), [ts0] "=&s"(ts0), [ts1] "=&s"(ts1)
        auto word = [&](const uint4 w) {
            if constexpr (R == 4) {
                asm volatile(OVM_WORD_ASM(OVM_ROWS_4, "")
                             : [p0] "+v"(pr[0]), [p1] "+v"(pr[1]), [p2] "+v"(pr[2]), [p3] "+v"(pr[3]), OVM_ACC_OPS(0, 0), OVM_ACC_OPS(0, 1),
                               OVM_ACC_OPS(0, 2), OVM_ACC_OPS(0, 3), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [b2] "=&v"(b[2]), [b3] "=&v"(b[3]),
                               OVM_WORD_TMP
                             : OVM_WORD_IN(w)
                             : "memory", "vcc", "scc");
            } else if constexpr (R == 2 && NOUT == 1) {
                asm volatile(OVM_WORD_ASM(OVM_ROWS_2, "")
                             : [p0] "+v"(pr[0]), [p1] "+v"(pr[1]), OVM_ACC_OPS(0, 0), OVM_ACC_OPS(0, 1), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]),
                               OVM_WORD_TMP
                             : OVM_WORD_IN(w)
                             : "memory", "vcc", "scc");
            } else if constexpr (R == 2) {
                asm volatile(OVM_WORD_ASM(OVM_ROWS_2, OVM_ACC1_ASM(OVM_ROWS_2))
                             : [p0] "+v"(pr[0]), [p1] "+v"(pr[1]), OVM_ACC_OPS(0, 0), OVM_ACC_OPS(0, 1), OVM_ACC_OPS(1, 0), OVM_ACC_OPS(1, 1),
                               [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), OVM_WORD_TMP
                             : OVM_WORD_IN(w)
                             : "memory", "vcc", "scc");
            } else if constexpr (NOUT == 1) {
                asm volatile(OVM_WORD_ASM(OVM_ROWS_1, "")
                             : [p0] "+v"(pr[0]), OVM_ACC_OPS(0, 0), [b0] "=&v"(b[0]), OVM_WORD_TMP
                             : OVM_WORD_IN(w)
                             : "memory", "vcc", "scc");
            } else {
                asm volatile(OVM_WORD_ASM(OVM_ROWS_1, OVM_ACC1_ASM(OVM_ROWS_1))
                             : [p0] "+v"(pr[0]), OVM_ACC_OPS(0, 0), OVM_ACC_OPS(1, 0), [b0] "=&v"(b[0]), OVM_WORD_TMP
                             : OVM_WORD_IN(w)
                             : "memory", "vcc", "scc");
            }
        };
#define OVM_G1(j) [g##j##0] "+v"(g[j][0])
#define OVM_G2(j) OVM_G1(j), [g##j##1] "+v"(g[j][1])
#define OVM_G4(j) OVM_G2(j), [g##j##2] "+v"(g[j][2]), [g##j##3] "+v"(g[j][3])
#define OVM_BLOCK_IN(W)                                                                                                     \
    [wx] "s"(W.x), [wy] "s"(W.y), [mine] "v"(mine), [recs] "s"(recs), [pow] "s"(pow), [ldelo] "s"(lde_lo), [ldehi] "s"(lde_hi), \
        [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3])
        auto block = [&](const uint4 w) {
            if constexpr (LEAVES) {
                asm volatile(OVM_BLOCK_LEAF_ASM(OVM_LAND_1)
                             : OVM_G1(0), OVM_G1(1), OVM_G1(2), OVM_G1(3), [ad] "=&v"(ad)
                             : OVM_BLOCK_IN(w)
                             : OVM_BLOCK_CLOBBERS);
            } else if constexpr (R == 4) {
                asm volatile(OVM_BLOCK_ASM(OVM_LAND_4, OVM_REQ_4)
                             : OVM_G4(0), OVM_G4(1), OVM_G4(2), OVM_G4(3), OVM_G4(4), OVM_G4(5), OVM_G4(6), OVM_G4(7), [ad] "=&v"(ad)
                             : OVM_BLOCK_IN(w)
                             : OVM_BLOCK_CLOBBERS);
            } else if constexpr (R == 2) {
                asm volatile(OVM_BLOCK_ASM(OVM_LAND_2, OVM_REQ_2)
                             : OVM_G2(0), OVM_G2(1), OVM_G2(2), OVM_G2(3), OVM_G2(4), OVM_G2(5), OVM_G2(6), OVM_G2(7), [ad] "=&v"(ad)
                             : OVM_BLOCK_IN(w)
                             : OVM_BLOCK_CLOBBERS);
            } else {
                asm volatile(OVM_BLOCK_ASM(OVM_LAND_1, OVM_REQ_1)
                             : OVM_G1(0), OVM_G1(1), OVM_G1(2), OVM_G1(3), OVM_G1(4), OVM_G1(5), OVM_G1(6), OVM_G1(7), [ad] "=&v"(ad)
                             : OVM_BLOCK_IN(w)
                             : OVM_BLOCK_CLOBBERS);
            }
        };
        // The words are fetched by the statements themselves, a group of four (one s_load_dwordx16) into A while B runs and
        // into B while A runs: a load the COMPILER issues is waited for right there, in front of the next asm statement.
        // The stream is a whole number of PAIRS of groups, plus one group that is fetched and never run.
        typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
        u32x16 A, B;
        auto run = [&](const u32x16 &q) {
            const uint4 w0 = make_uint4(q[0], q[1], q[2], q[3]), w1 = make_uint4(q[4], q[5], q[6], q[7]);
            const uint4 w2 = make_uint4(q[8], q[9], q[10], q[11]), w3 = make_uint4(q[12], q[13], q[14], q[15]);
            block(w0);
            word(w0);
            word(w1);
            word(w2);
            word(w3);
            if constexpr (LEAVES) {
                // a leaf marker is the LAST word of its group: emit, then clear accumulator 0 -- in an asm statement of its
                // own with an internal skip, like every other write to the accumulators (no phi, no copies)
                if ((w3.x >> 29) == 3u) leaf(w3.y, Fp::raw(pr[0]), acc0[0]);
                asm volatile("s_lshr_b32 %[ts0], %[wx], 29\n\ts_cmp_lg_u32 %[ts0], 3\n\ts_cbranch_scc1 1f\n\t"
                             "v_mov_b32 %[h0], 0\n\tv_mov_b32 %[h1], 0\n\tv_mov_b32 %[h2], 0\n\tv_mov_b32 %[h3], 0\n\t"
                             "v_mov_b64 %[l0], 0\n\tv_mov_b64 %[l1], 0\n\tv_mov_b64 %[l2], 0\n\tv_mov_b64 %[l3], 0\n"
                             "1:"
                             : [l0] "+v"(acc0[0].lo[0]), [l1] "+v"(acc0[0].lo[1]), [l2] "+v"(acc0[0].lo[2]), [l3] "+v"(acc0[0].lo[3]),
                               [h0] "+v"(acc0[0].hi[0]), [h1] "+v"(acc0[0].hi[1]), [h2] "+v"(acc0[0].hi[2]), [h3] "+v"(acc0[0].hi[3]),
                               [ts0] "=&s"(ts0)
                             : [wx] "s"(w3.x)
                             : "scc");
            }
        };
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(A) : "s"(wp) : "memory");
        for (uint32_t i = n_words; i != 0; i -= 8) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %0, %1, 0x40" : "=s"(B) : "s"(wp) : "memory");
            run(A);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %0, %1, 0x80" : "=s"(A) : "s"(wp) : "memory");
            run(B);
            wp += 8;
        }
        // (the last block word requests nothing: no column load is in flight; the group fetched past the end and the stores to
        // LDS are waited for here)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef OVM_ACC_OPS
#undef OVM_WORD_IN
#undef OVM_WORD_TMP
#undef OVM_G1
#undef OVM_G2
#undef OVM_G4
#undef OVM_BLOCK_IN
#endif
    }
};

// Round 0: the program's accumulators at the lane's R rows of coset k (eval.cuh: round0_eval_body).
template <int R, int NOUT>
struct CachedRound0 {
    static constexpr int ROWS = R;
    static constexpr bool REDUCE_IN_IMAGE = true;  // (eval.cuh: the launch gives the image at least RED_THREADS * 16 bytes)
    DevCached prog;
    __device__ __forceinline__ void operator()(const Fp *lde_k, uint32_t log_lifted, const uint64_t (&p)[R],
                                               EF (&out)[R][2]) const {
        uint32_t o[4];  // byte offset of the lane's r-th cell inside a column
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = r < R ? (uint32_t)p[r < R ? r : 0] << 2 : 0;
        Acc96 acc0[R], acc1[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc0[r].clear(), acc1[r].clear();
        CachedCore<R, NOUT, false>::run(prog, (uint64_t)(uintptr_t)lde_k, 1u << (log_lifted + 2) /* bytes per column (log_lifted <= 27) */, o,
                                        acc0, acc1, [](uint32_t, Fp, const Acc96 &) {});
#pragma unroll
        for (int r = 0; r < R; r++) {
            out[r][0] = out[r][0] + acc0[r].finish();
            if (NOUT > 1) out[r][1] = out[r][1] + acc1[r].finish();
        }
    }
};

}  // namespace ovm
