// Switches of the prover read from the environment, in ONE place (host only).
//
// The PRODUCT library reads the deployer's switches and nothing else (INTEGRATION.md "Environment"; the reference's CUDA
// backend likewise exposes three documented prover switches, crates/cuda-backend/src/device.rs:25-30, and one tuning family,
// logup_zerocheck/fractional.rs:209-243):
//     OVM_TRACE                 wall-clock trace of the stages on stderr
//     OVM_ROCTX                 the stage labels as roctx ranges (runtime.hip)
//     OVM_SHARD_TIMEOUT_S       a sharded proof gives up on a silent peer after this long (shard.cpp)
//     OVM_IPC_OPEN_TIMEOUT_S    ... and on a peer's exported buffer that does not map (ipc_comm.cpp)
// Everything else below -- layouts, alternative kernels the parity tests force on small traces, fault injection -- exists only
// in the TEST TWIN of the library (-DOVM_TEST_HOOKS=1, tools/build_hooks.sh), which the GPU suite loads through OVM_HIP_LIB
// next to the product: in the product these fields are compile-time defaults and their environment names are not even
// strings of the binary.
#pragma once
#include <cstdint>

namespace ovm {

struct Tuning {
    bool trace = false;            // OVM_TRACE
    // ---- test hooks (-DOVM_TEST_HOOKS=1 only) ----
    int r0_rows = 0;               // OVM_R0_ROWS=1|2|4: trace rows per lane of round 0 (0: by height)
    int stage_rows = 0;            // OVM_STAGE_ROWS: LDS rows of the round-0 image / of one staging buffer of the leaf kernel (0: policy)
    int acc_rows = 0;              // OVM_ACC_ROWS: columns a round-0 block may bring / accumulation rows of the leaf kernel (0: policy)
    uint32_t stage_kib = 0;        // OVM_STAGE_KIB: LDS image budget per workgroup (0: policy)
    bool mle_lds_slots = true;     // OVM_MLE_LDS=0: the large multilinear rounds keep their slots in private memory (the fallback of programs with many slots)
    bool mle_forward = true;       // OVM_MLE_FWD=0: the LDS-slot kernel ignores the compiler's forwarding hints
    bool r0_batch = true;          // OVM_R0_BATCH=0: every trace's round 0 its own launch
    int r0_groups_min = -1;        // OVM_R0_GROUPS_MIN: programs of at least this many groups run grouped (-1: default)
    uint32_t mle_batch_wgs = 256;  // OVM_MLE_BATCH: jobs up to this many workgroups share a launch (0: none)
    uint32_t mle_groups_max = 128; // OVM_MLE_GROUPS_MAX
    bool gkr_three_points = false; // OVM_GKR_THREE_POINTS: every GKR sub-round takes the three-point fallback
    bool merkle_coop = true;       // OVM_MERKLE_COOP=0: no 16-lane cooperative compress for the small layers
    uint32_t finish_inline_max = 64;  // OVM_FINISH_INLINE_MAX: grids up to this many workgroups finish their reduction in the
                                   // kernel (last workgroup sums the records); larger ones take a second launch
    bool early_lde = true;         // OVM_EARLY_LDE=0: round 0's coset LDE runs where it is used, not beside the GKR layer sumchecks
    // OVM_EARLY_LDE_LDS_KIB: LDS each workgroup of the early LDE claims without using it. 64 KiB = two workgroups per
    // CU: the sumcheck kernels of the caller's stream find free wave slots at once (profiles/r03/early_lde_sweep.txt)
    uint32_t early_lde_lds_kib = 64;
    // OVM_MLE_BATCH_LDS_MIN_LOG: batched multilinear-round jobs of at least 2^this pairs run on the LDS-slot interpreter
    // (all points of a pair in the lane); shorter ones keep one (pair, point) per lane. 31 = never.
    uint32_t mle_batch_lds_min_log = 9;
    bool lde_transposed = true;    // OVM_LDE_XPOSE=0: the coset LDE of tall traces moves its chunks lane by lane, not through LDS
    // OVM_STACKED_TAIL_DEVICE: the finished slices of a stacked-reduction round are summed 0 = by the host, 1 = on the device
    // when there are at least 1 024 of them (a many-AIR proof), 2 = always on the device (parity tests)
    int stacked_tail_device = 1;
    bool gkr_split_eq = true;      // OVM_GKR_SPLIT_EQ=0: every GKR layer builds and reads its whole eq table
    uint32_t gkr_split_min_round = 14;  // OVM_GKR_SPLIT_MIN: layers of at least 2^this records use split tables (>= 3; parity tests lower it)
    bool lazy_w_accumulate = true;  // OVM_LAZY_W=0: whir_w_accumulate reduces every product (the parity tests compare the two)
    bool download_kernel = true;   // OVM_DOWNLOAD_KERNEL=0: small downloads through the runtime's copy + stream synchronisation
    int test_fail_rank = -1;       // OVM_TEST_FAIL_RANK: this rank of a sharded proof throws after its commitment
    static Tuning from_env();
    // The environment is read ONCE per thread (first use). Test processes set OVM_TUNING_LIVE=1 (tests/conftest.py, the
    // fuzzers) and load the test twin: then every prover entry re-reads it, so a test can switch a hook between two proofs.
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

#if defined(OVM_TEST_HOOKS)
inline constexpr bool TEST_HOOKS = true;
#else
inline constexpr bool TEST_HOOKS = false;
#endif

}  // namespace ovm
