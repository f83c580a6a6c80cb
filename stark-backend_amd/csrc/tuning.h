// Tuning / test overrides of the prover, read from the environment in ONE place (host only).
#pragma once
#include <cstdint>

namespace ovm {

// Every tuning / test override the prover honours, in one place. Production callers leave the environment
// alone and get the defaults; the parity tests force the alternative kernels and layouts on small traces.
// Read ONCE per call of a prover entry point (Tuning::refresh in guarded(), prover.hip), never inside a stage;
// the kernel-level entries (ovm_hip_merkle_tree ...) use what the calling thread read first.
struct Tuning {
    int r0_rows = 0;               // OVM_R0_ROWS=1|2|4: trace rows per lane of the staged round 0 (0: by height)
    int stage_rows = 0;            // OVM_STAGE_ROWS: LDS rows of one staging buffer (0: from the budget)
    int acc_rows = 0;              // OVM_ACC_ROWS: LDS accumulation rows (0: from the budget)
    uint32_t stage_kib = 0;        // OVM_STAGE_KIB: LDS image budget per workgroup (0: 36 KiB, 72 for long programs on tall traces)
    bool mle_points_in_lane = true;  // OVM_MLE_POINTS=0: one evaluation point per lane in the large rounds too
    bool mle_lds_slots = true;     // OVM_MLE_LDS=0: the large multilinear rounds keep their slots in private memory (rounds 1-2)
    bool mle_forward = true;       // OVM_MLE_FWD=0: the LDS-slot kernel ignores the compiler's forwarding hints
    bool r0_batch = true;          // OVM_R0_BATCH=0: every trace's round 0 its own launch
    int r0_groups_min = -1;        // OVM_R0_GROUPS_MIN: programs of at least this many groups run grouped (-1: default)
    uint32_t mle_batch_wgs = 256;  // OVM_MLE_BATCH: jobs up to this many workgroups share a launch (0: none)
    uint32_t mle_groups_max = 128; // OVM_MLE_GROUPS_MAX
    bool gkr_three_points = false; // OVM_GKR_THREE_POINTS: every GKR sub-round takes the three-point fallback
    bool merkle_coop = true;       // OVM_MERKLE_COOP=0: no 16-lane cooperative compress for the small layers
    uint32_t finish_inline_max = 64;    // OVM_FINISH_INLINE_MAX: grids up to this many workgroups finish their reduction
                                   // in the kernel (last workgroup sums the records); larger ones take a second launch
    uint32_t finish_atomic_max = 0;     // OVM_FINISH_ATOMIC_MAX: grids above finish_inline_max and up to this many workgroups
                                   // finish in the kernel through 64-bit atomic accumulators. 0 (default): never --
                                   // measured no faster in the prover than the second launch (stages.hip)
    bool early_lde = true;         // OVM_EARLY_LDE=0: round 0's coset LDE runs where it is used, not beside the GKR layer sumchecks
    // OVM_EARLY_LDE_LDS_KIB: LDS each workgroup of the early LDE claims without using it. 64 KiB = two workgroups per
    // CU: the sumcheck kernels of the caller's stream find free wave slots at once (0 / 32 / 48 / 64 KiB: 75.8 / 75.3 /
    // 75.2 / 74.8 ms per proof, 75.8 ms without the early LDE; profiles/r03/early_lde_sweep.txt)
    uint32_t early_lde_lds_kib = 64;
    // OVM_MLE_BATCH_LDS_MIN_LOG: batched multilinear-round jobs of at least 2^this pairs run on the LDS-slot interpreter
    // (all points of a pair in the lane); shorter ones keep one (pair, point) per lane. 31 = never.
    uint32_t mle_batch_lds_min_log = 9;  // (a 46-AIR reth segment: 7 / 8 / 9 / 10 -> 19.0 / 17.2 / 16.3 / 16.9 ms of multilinear rounds, 22.3 without)
    // OVM_STACKED_TAIL_DEVICE: the finished slices of a stacked-reduction round are summed 0 = by the host, 1 = on the device
    // when there are at least 1 024 of them (a many-AIR proof), 2 = always on the device (parity tests)
    int stacked_tail_device = 1;
    bool staged_forward = true;    // OVM_STAGED_FWD=0: the staged VM reads every operand from LDS and stores every result
    bool gkr_split_eq = true;      // OVM_GKR_SPLIT_EQ=0: every GKR layer builds and reads its whole eq table
    int gkr_build_fused = 3;       // OVM_GKR_BUILD_FUSED=0|2|3: levels of the fraction tree built per pass (0 = 1; stages.hip gkr_build2/3_kernel)
    uint32_t gkr_split_min_round = 14;  // OVM_GKR_SPLIT_MIN: layers of at least 2^this records use split tables (>= 3; parity tests lower it)
    // OVM_GKR_LOOKAHEAD=1: a pass over a GKR layer's table also sums the NEXT sub-round as a quadratic in its challenge, so
    // every second sub-round costs no pass and no launch (stages.hip: gkr_pass_kernel). Built and byte-exact in round 4, and
    // OFF by default: measured no faster at any size limit (profiles/r04/gkr_lookahead_sweep/).
    bool gkr_lookahead = false;
    // OVM_GKR_LA_MAX_LOG: only passes over at most 2^this pairs carry a lookahead. The six extra sums make a pass over a BIG
    // table ALU-bound (2.6x the extension-field products) and its double fold reads 256 B per lane and plane (3.2x the HBM
    // fetch, measured: profiles/r04/gkr_lookahead_all_layers_*.txt): +2.7 ms of kernel time at config D for 100 round trips
    // saved. The SMALL sub-rounds are the latency-bound ones -- a table that sits in L2, a kernel of a few microseconds -- and
    // there the saved launch + mailbox round trip is all gain.
    uint32_t gkr_la_max_log = 17;
    // OVM_GKR_LA_MIN_LOG: ... and only passes over at least 2^this pairs (0 = no lower limit)
    uint32_t gkr_la_min_log = 0;
    bool lazy_w_accumulate = true;  // OVM_LAZY_W=0: whir_w_accumulate reduces every product (the parity tests compare the two)
    bool download_kernel = true;   // OVM_DOWNLOAD_KERNEL=0: small downloads through the runtime's copy + stream synchronisation
    bool trace = false;            // OVM_TRACE: wall-clock trace of the stages on stderr
    int test_fail_rank = -1;       // OVM_TEST_FAIL_RANK: this rank of a sharded proof throws after its commitment (tests)
    static Tuning from_env();
    // The environment is read ONCE per thread (first use). Test processes set OVM_TUNING_LIVE=1 (tests/conftest.py, the
    // fuzzers): then every prover entry re-reads it, so a test can switch a knob between two proofs. Production callers
    // never pay for ~40 getenv calls per proof, and a knob exported by accident after start-up changes nothing mid-run.
    static bool live();
    static void refresh() {
        if (live()) current() = from_env();
    }
    static Tuning &current() {
        static thread_local Tuning t = from_env();
        return t;
    }
};
inline const Tuning &tuning() { return Tuning::current(); }

}  // namespace ovm
