//! Raw bindings of `include/openvm_hip.h` (libopenvm_hip.so): one declaration per C entry, same
//! order, same argument order. Peer of crates/cuda-backend/src/cuda/*.rs. `tests/test_abi.py` in the
//! HIP backend's repository parses this file and the header and compares every declaration (name,
//! argument count, each argument's and the return value's type), as scripts/gen_stub.py +
//! scripts/abi_cmp.py do for the CUDA backend; keep the two in step.
#![allow(non_camel_case_types, clippy::too_many_arguments)]
use std::os::raw::{c_char, c_int, c_void};

/// `ovm_stream_t` (= `hipStream_t`; null = default stream)
pub type OvmStream = *mut c_void;

/// `ovm_sponge_t`: the transcript's duplex sponge (reference: DeviceSpongeState, cuda-backend/src/sponge.rs:39-48)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct OvmSponge {
    pub state: [u32; 16],
    pub absorb_idx: u32,
    pub sample_idx: u32,
}

/// `ovm_pk_t`, `ovm_pcs_t`, `ovm_shm_comm_t`: opaque handles
#[repr(C)]
pub struct OvmPk {
    _private: [u8; 0],
}
#[repr(C)]
pub struct OvmPcs {
    _private: [u8; 0],
}
#[repr(C)]
pub struct OvmShmComm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct OvmIpcComm {
    _private: [u8; 0],
}

/// `ovm_ipc_status_t`: which carrier the library's IPC exchange is on (mapped peer buffers, or -- after a failed or overdue
/// `hipIpcOpenMemHandle` -- staged through host memory) and why.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct OvmIpcStatus {
    pub degraded: u32,
    pub mappings: u32,
    pub opens: u64,
    pub exchanges: u64,
    pub staged_exchanges: u64,
    pub bounced_exchanges: u64,
    pub reason: [c_char; 192],
}

/// `ovm_trace_t` (reference: AirProvingContext, prover/types.rs:59-73)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct OvmTrace {
    pub air_idx: u32,
    pub num_public_values: u32,
    pub d_common_main: *const u32,
    pub height: u64,
    pub public_values: *const u32,
    pub num_cached: u32,
    pub _pad: u32,
    pub cached: *const *mut OvmPcs,
}

/// `ovm_pcs_params_t`
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct OvmPcsParams {
    pub l_skip: u32,
    pub n_stack: u32,
    pub log_blowup: u32,
    pub k_whir: u32,
}

/// `ovm_matrix_t`
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct OvmMatrix {
    pub d_ptr: *const u32,
    pub width: u64,
    pub height: u64,
}

/// `ovm_prover_config_t` (reference: GpuProverConfig, cuda-backend/src/device.rs:25-30)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct OvmProverConfig {
    pub cache_rs_code_matrix: u32,
    pub zerocheck_save_memory: u32,
    pub round0_slab_log_rows: u32,
    pub gkr_drop_leaves: u32,
    pub cache_limit_bytes: u64,
}

/// `ovm_shard_t`: the communicator of a sharded proof
#[repr(C)]
#[derive(Clone, Copy)]
pub struct OvmShard {
    pub rank: u32,
    pub world: u32,
    pub ctx: *mut c_void,
    pub allgather_host:
        Option<unsafe extern "C" fn(ctx: *mut c_void, mine: *const c_void, all: *mut c_void, bytes: u64) -> c_int>,
    pub allgather_device: Option<
        unsafe extern "C" fn(ctx: *mut c_void, d_mine: *const c_void, d_all: *mut c_void, bytes: u64, stream: OvmStream) -> c_int,
    >,
    /// optional (`None`: the mu-batch vectors are summed through an all-gather of whole vectors)
    pub alltoall_device: Option<
        unsafe extern "C" fn(ctx: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, bytes: u64, stream: OvmStream) -> c_int,
    >,
    /// optional: "this rank failed in the middle of a joint call" (built in: `ovm_hip_shm_comm_abort` / `ovm_hip_ipc_abort`)
    pub abort: Option<unsafe extern "C" fn(ctx: *mut c_void, code: c_int) -> c_int>,
}
pub const OVM_ESHARD_TIMEOUT: c_int = 10006;
pub const OVM_ESHARD_PEER_ABORTED: c_int = 10007;
pub const OVM_ESHARD_POISONED: c_int = 10008;

extern "C" {
    pub fn ovm_hip_abi_version() -> u32;
    pub fn ovm_hip_init(device: c_int, cu_count: *mut u32) -> c_int;
    pub fn ovm_hip_thread_shutdown() -> c_int;
    pub fn ovm_hip_error_name(code: c_int) -> *const c_char;
    pub fn ovm_hip_profile_enable(on: c_int) -> c_int;
    pub fn ovm_hip_profile_report(buf: *mut c_char, cap: u64) -> u64;
    pub fn ovm_hip_malloc(d_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn ovm_hip_free(d_ptr: *mut c_void) -> c_int;
    pub fn ovm_hip_memcpy_h2d(
        d_dst: *mut c_void,
        src: *const c_void,
        bytes: usize,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_memcpy_d2h(
        dst: *mut c_void,
        d_src: *const c_void,
        bytes: usize,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_memcpy_d2d(
        d_dst: *mut c_void,
        d_src: *const c_void,
        bytes: usize,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_memset(d_dst: *mut c_void, value: c_int, bytes: usize, stream: OvmStream) -> c_int;
    pub fn ovm_hip_stream_create(stream: *mut OvmStream) -> c_int;
    pub fn ovm_hip_stream_destroy(stream: OvmStream) -> c_int;
    pub fn ovm_hip_stream_synchronize(stream: OvmStream) -> c_int;
    pub fn ovm_hip_eval_to_coeff(
        d_in: *const u32,
        d_out: *mut u32,
        width: u64,
        log_height: u32,
        l_skip: u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_ntt_dif(
        d_in: *const u32,
        d_out: *mut u32,
        width: u64,
        log_n: u32,
        log_n_in: u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_ntt_dif_ext(
        d_in: *const u32,
        d_out: *mut u32,
        width: u64,
        log_n: u32,
        log_n_in: u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_rs_encode(
        d_stacked: *const u32,
        d_scratch: *mut u32,
        d_codeword: *mut u32,
        width: u64,
        log_height: u32,
        l_skip: u32,
        log_blowup: u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_merkle_scratch_words(log_rows: u32, log_rows_per_query: u32) -> u64;
    pub fn ovm_hip_merkle_layers_words(log_rows: u32, log_rows_per_query: u32) -> u64;
    pub fn ovm_hip_merkle_tree(
        d_matrix: *const u32,
        width: u64,
        log_rows: u32,
        log_rows_per_query: u32,
        d_scratch: *mut u32,
        d_digest_layers: *mut u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_merkle_tree_ext(
        d_matrix: *const u32,
        width: u64,
        log_rows: u32,
        log_rows_per_query: u32,
        d_scratch: *mut u32,
        d_digest_layers: *mut u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_poseidon2_permute(d_states: *mut u32, n: u64, stream: OvmStream) -> c_int;
    pub fn ovm_hip_poseidon2_permute_rc(d_states: *mut u32, n: u64, d_rc141: *const u32, stream: OvmStream) -> c_int;
    pub fn ovm_hip_sponge_grind(
        sponge: *const OvmSponge,
        bits: u32,
        witness_out: *mut u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_gather_rows(
        d_matrix: *const u32,
        width: u64,
        col_stride: u64,
        words_per_elem: u32,
        d_row_idx: *const u64,
        n_rows: u64,
        d_out: *mut u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_gather_digests(
        d_layers: *const u32,
        d_idx: *const u64,
        n: u64,
        d_out: *mut u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_pk_load(blob: *const u32, words: u64, pk_out: *mut *mut OvmPk, stream: OvmStream) -> c_int;
    pub fn ovm_hip_pk_free(pk: *mut OvmPk);
    pub fn ovm_hip_pk_preprocessed_commitment(
        pk: *const OvmPk,
        air_idx: u32,
        commitment_out: *mut u32,
    ) -> c_int;
    pub fn ovm_hip_pk_preprocessed_data(
        pk: *const OvmPk,
        air_idx: u32,
        d_trace_out: *mut *const u32,
        width_out: *mut u64,
        height_out: *mut u64,
        commitment_out: *mut u32,
        pcs_out: *mut *mut OvmPcs,
    ) -> c_int;
    pub fn ovm_hip_program_stats(
        blob: *const u32,
        words: u64,
        air_idx: u32,
        kind: u32,
        instrs_out: *mut u32,
        slots_out: *mut u32,
        remat_limit_out: *mut u32,
    ) -> c_int;
    pub fn ovm_hip_staged_selftest(
        blob: *const u32,
        words: u64,
        stage_rows: u32,
        acc_rows: u32,
        seed: u64,
        blocks_out: *mut u64,
    ) -> c_int;
    pub fn ovm_hip_host_gkr_batch_selftest(rounds: u32, seed: u64) -> c_int;
    pub fn ovm_hip_host_permute_selftest(n: u32, seed: u64) -> c_int;
    pub fn ovm_hip_set_prover_config(cfg: *const OvmProverConfig) -> c_int;
    pub fn ovm_hip_memory_stats(
        live_bytes: *mut u64,
        peak_live_bytes: *mut u64,
        cached_bytes: *mut u64,
        reset_peak: c_int,
    ) -> c_int;
    pub fn ovm_hip_pool_stats(out4: *mut u64) -> c_int;
    pub fn ovm_hip_pool_selftest(seed: u64, ops: u32) -> c_int;
    pub fn ovm_hip_constraint_eval_buffer_size(pk: *const OvmPk, air_idx: u32) -> u64;
    pub fn ovm_hip_trim() -> c_int;
    pub fn ovm_hip_last_error() -> *const c_char;
    pub fn ovm_hip_commit(
        pk: *const OvmPk,
        n_traces: u32,
        traces: *const OvmTrace,
        commitment_out: *mut u32,
        pcs_out: *mut *mut OvmPcs,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_commit_matrix(
        pk: *const OvmPk,
        d_matrix: *const u32,
        width: u64,
        height: u64,
        commitment_out: *mut u32,
        pcs_out: *mut *mut OvmPcs,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_commit_matrices(
        params: *const OvmPcsParams,
        n: u32,
        mats: *const OvmMatrix,
        commitment_out: *mut u32,
        pcs_out: *mut *mut OvmPcs,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_proof_size(
        pk: *const OvmPk,
        n_traces: u32,
        traces: *const OvmTrace,
        section_ends5: *mut u64,
    ) -> c_int;
    pub fn ovm_hip_pcs_free(pcs: *mut OvmPcs);
    pub fn ovm_hip_prove_rap_constraints(
        pk: *const OvmPk,
        n_traces: u32,
        traces: *const OvmTrace,
        sponge: *mut OvmSponge,
        out: *mut u8,
        out_cap: u64,
        gkr_len: *mut u64,
        total_len: *mut u64,
        r_out: *mut u32,
        r_cap: u32,
        r_len: *mut u32,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_prove_openings(
        pk: *const OvmPk,
        n_traces: u32,
        traces: *const OvmTrace,
        common_main: *mut OvmPcs,
        r: *const u32,
        r_len: u32,
        sponge: *mut OvmSponge,
        out: *mut u8,
        out_cap: u64,
        stacking_len: *mut u64,
        total_len: *mut u64,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_prove(
        pk: *const OvmPk,
        n_traces: u32,
        traces: *const OvmTrace,
        sponge: *mut OvmSponge,
        proof_out: *mut u8,
        proof_cap: u64,
        proof_len: *mut u64,
        section_offsets5: *mut u64,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_shm_comm_create(
        name: *const c_char,
        rank: u32,
        world: u32,
        slot_bytes: u64,
        out: *mut *mut OvmShmComm,
    ) -> c_int;
    pub fn ovm_hip_shm_segment_bytes(world: u32, slot_bytes: u64) -> u64;
    pub fn ovm_hip_shm_comm_attach(
        base: *mut c_void,
        rank: u32,
        world: u32,
        slot_bytes: u64,
        out: *mut *mut OvmShmComm,
    ) -> c_int;
    pub fn ovm_hip_shm_comm_destroy(comm: *mut OvmShmComm);
    pub fn ovm_hip_shm_allgather(
        comm: *mut c_void,
        mine: *const c_void,
        all: *mut c_void,
        bytes: u64,
    ) -> c_int;
    pub fn ovm_hip_shm_comm_abort(comm: *mut c_void, code: c_int) -> c_int;
    pub fn ovm_hip_shm_comm_poisoned(comm: *const OvmShmComm) -> c_int;
    pub fn ovm_hip_shard_last_error() -> *const c_char;
    pub fn ovm_hip_ipc_abort(comm: *mut c_void, code: c_int) -> c_int;
    pub fn ovm_hip_gkr_shard_selftest(
        world: u32,
        tr: u32,
        ends: *const u64,
        seed: u64,
        three_points: c_int,
    ) -> c_int;
    pub fn ovm_hip_selftest_error() -> *const c_char;
    pub fn ovm_hip_ipc_comm_create(shm: *mut OvmShmComm, rank: u32, world: u32, out: *mut *mut OvmIpcComm) -> c_int;
    pub fn ovm_hip_ipc_comm_destroy(comm: *mut OvmIpcComm);
    pub fn ovm_hip_ipc_allgather_host(comm: *mut c_void, mine: *const c_void, all: *mut c_void, bytes: u64) -> c_int;
    pub fn ovm_hip_ipc_allgather_device(
        comm: *mut c_void,
        d_mine: *const c_void,
        d_all: *mut c_void,
        bytes: u64,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_ipc_alltoall_device(
        comm: *mut c_void,
        d_send: *const c_void,
        d_recv: *mut c_void,
        bytes: u64,
        stream: OvmStream,
    ) -> c_int;
    pub fn ovm_hip_ipc_comm_status(comm: *const OvmIpcComm, out: *mut OvmIpcStatus) -> c_int;
    pub fn ovm_hip_shard_exchange_selftest(shard: *const OvmShard, n: u64, rounds: u32, seed: u64) -> c_int;
    pub fn ovm_hip_prove_sharded(
        pk: *const OvmPk,
        shard: *const OvmShard,
        n_traces: u32,
        traces: *const OvmTrace,
        owner: *const u32,
        sponge: *mut OvmSponge,
        proof_out: *mut u8,
        proof_cap: u64,
        proof_len: *mut u64,
        section_offsets5: *mut u64,
        stream: OvmStream,
    ) -> c_int;
}
