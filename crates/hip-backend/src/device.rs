//! `HipDevice`: the three prover stages of hal.rs:84-138, one C-ABI call each. Peer of
//! crates/cuda-backend/src/{device.rs,gpu_backend.rs}; the stage drivers themselves
//! (stacked_pcs.rs, logup_zerocheck/, stacked_reduction.rs, whir.rs there) live behind the ABI here.
use std::{ptr, sync::Arc};

use openvm_stark_backend::{
    codec::Decode,
    proof::{BatchConstraintProof, GkrProof, StackingProof, WhirProof},
    prover::{
        DeviceMultiStarkProvingKey, DeviceStarkProvingKey, MatrixDimensions, MultiRapProver, OpeningProver, ProverBackend,
        ProverDevice, ProvingContext, TraceCommitter,
    },
    SystemParams,
};

use crate::{
    base::{AirDataHip, HipMatrix, StackedPcsDataHip},
    error::{CommitError, HipError, OpeningError, ProverError, RapConstraintsError},
    ffi,
    prelude::{Digest, D_EF, EF, F, SC},
    transcript::DuplexSpongeHip,
};

/// `ProverBackend` for BabyBear + Poseidon2 on MI355X (hal.rs:23-63).
#[derive(Clone, Copy, Default)]
pub struct HipBackend;

impl ProverBackend for HipBackend {
    const CHALLENGE_EXT_DEGREE: u8 = D_EF as u8;
    type Val = F;
    type Challenge = EF;
    type Commitment = Digest;
    type Matrix = HipMatrix<F>;
    type PcsData = StackedPcsDataHip;
    type OtherAirData = AirDataHip;

    /// hal.rs:57-62: bytes of round-0 scratch per trace row of this AIR -- the coset LDE of its view
    /// columns (the constraint VM itself keeps its intermediates in LDS). With
    /// `zerocheck_save_memory` the library multiplies it by the slab, not by the trace height.
    fn constraint_eval_buffer_size(pk: &DeviceStarkProvingKey<Self>) -> usize {
        // SAFETY: the key handle is valid for the life of the proving key; host-only query.
        unsafe { ffi::ovm_hip_constraint_eval_buffer_size(pk.other_data.key.0, pk.other_data.air_idx) as usize }
    }
}

/// Memory-bounded modes (peer of `GpuProverConfig`, cuda-backend/src/device.rs:25-30). The defaults keep
/// everything resident -- 288 GB of HBM is the normal case; proofs are identical in every mode.
/// `cache_stacked_matrix` has no counterpart: column-major traces stack by concatenation and a trace
/// that fills the stacked matrix is never copied.
#[derive(Clone, Copy, Debug)]
pub struct HipProverConfig {
    /// `false`: a commitment drops its codeword once the Merkle tree is built; WHIR re-encodes it when
    /// it opens rows (cuda-backend/src/whir.rs:404-416).
    pub cache_rs_code_matrix: bool,
    /// `true`: round 0 extends and evaluates a tall trace in slabs of `2^round0_slab_log_rows` rows.
    pub zerocheck_save_memory: bool,
    pub round0_slab_log_rows: u32,
    /// `true`: the leaf layer of the LogUp fraction tree is dropped after the first tree layer is built
    /// and generated again for the last layer's sumcheck (the CUDA backend's reversible in-place tree,
    /// docs/cuda-backend/gkr-prover.md:129-133, answers the same need).
    pub gkr_drop_leaves: bool,
    /// Upper bound on what the thread's caching allocator keeps in free blocks (0: unbounded).
    pub cache_limit_bytes: u64,
}
impl Default for HipProverConfig {
    fn default() -> Self {
        Self {
            cache_rs_code_matrix: true,
            zerocheck_save_memory: false,
            round0_slab_log_rows: 20,
            gkr_drop_leaves: false,
            cache_limit_bytes: 0,
        }
    }
}
impl HipProverConfig {
    /// Applies to the provers of the CALLING thread (one engine per OS thread).
    pub fn apply(&self) -> Result<(), HipError> {
        let cfg = ffi::OvmProverConfig {
            cache_rs_code_matrix: self.cache_rs_code_matrix as u32,
            zerocheck_save_memory: self.zerocheck_save_memory as u32,
            round0_slab_log_rows: self.round0_slab_log_rows,
            gkr_drop_leaves: self.gkr_drop_leaves as u32,
            cache_limit_bytes: self.cache_limit_bytes,
        };
        // SAFETY: plain struct, read during the call only.
        HipError::from_result("ovm_hip_set_prover_config", unsafe { ffi::ovm_hip_set_prover_config(&cfg) })
    }
}

/// Stream-owning device handle (cuda-backend/src/device.rs:49-58). `Clone` shares the stream: one
/// engine per OS thread is the supported concurrency (cuda-backend/examples/keccakf.rs:88-131).
#[derive(Clone)]
pub struct HipDevice {
    params: SystemParams,
    pub id: u32,
    pub cu_count: u32,
    ctx: HipDeviceCtx,
}

/// `ProverDevice::DeviceCtx`: the HIP stream all of a device's work is enqueued on.
#[derive(Clone)]
pub struct HipDeviceCtx {
    stream: Arc<StreamHandle>,
}
struct StreamHandle(ffi::OvmStream);
// SAFETY: a HIP stream handle may be used from any thread.
unsafe impl Send for StreamHandle {}
unsafe impl Sync for StreamHandle {}
impl Drop for StreamHandle {
    fn drop(&mut self) {
        // SAFETY: created by ovm_hip_stream_create, destroyed once.
        unsafe { ffi::ovm_hip_stream_destroy(self.0) };
    }
}
impl HipDeviceCtx {
    pub fn stream(&self) -> ffi::OvmStream {
        self.stream.0
    }
    pub fn synchronize(&self) -> Result<(), HipError> {
        // SAFETY: valid stream.
        HipError::from_result("ovm_hip_stream_synchronize", unsafe { ffi::ovm_hip_stream_synchronize(self.stream.0) })
    }
}

impl HipDevice {
    pub fn new(params: SystemParams) -> Result<Self, HipError> {
        Self::for_device(0, params)
    }
    pub fn for_device(id: u32, params: SystemParams) -> Result<Self, HipError> {
        let mut cu_count = 0u32;
        // SAFETY: out-pointer is valid; builds the twiddle tables for this device on first use.
        HipError::from_result("ovm_hip_init", unsafe { ffi::ovm_hip_init(id as i32, &mut cu_count) })?;
        let mut stream: ffi::OvmStream = ptr::null_mut();
        HipError::from_result("ovm_hip_stream_create", unsafe { ffi::ovm_hip_stream_create(&mut stream) })?;
        Ok(Self { params, id, cu_count, ctx: HipDeviceCtx { stream: Arc::new(StreamHandle(stream)) } })
    }
    pub fn params(&self) -> &SystemParams {
        &self.params
    }
    pub fn stream(&self) -> ffi::OvmStream {
        self.ctx.stream()
    }
}

/// `ProvingContext` -> the POD trace list of the C ABI. The returned vectors own what the
/// `OvmTrace` entries point into and must outlive the call.
pub(crate) struct AbiTraces {
    pub traces: Vec<ffi::OvmTrace>,
    _public_values: Vec<Vec<u32>>,
    _cached: Vec<Vec<*mut ffi::OvmPcs>>,
}
pub(crate) fn abi_traces(ctx: &ProvingContext<HipBackend>) -> AbiTraces {
    let mut pvs = Vec::with_capacity(ctx.per_trace.len());
    let mut cached = Vec::with_capacity(ctx.per_trace.len());
    for (_, air_ctx) in &ctx.per_trace {
        // `BabyBear` is `repr(transparent)` over its Montgomery word: the words cross as they are
        pvs.push(air_ctx.public_values.iter().map(|x| unsafe { std::mem::transmute::<F, u32>(*x) }).collect::<Vec<u32>>());
        cached.push(air_ctx.cached_mains.iter().map(|c| c.data.handle).collect::<Vec<_>>());
    }
    let traces = ctx
        .per_trace
        .iter()
        .enumerate()
        .map(|(i, (air_idx, air_ctx))| ffi::OvmTrace {
            air_idx: *air_idx as u32,
            num_public_values: pvs[i].len() as u32,
            d_common_main: air_ctx.common_main.as_ptr() as *const u32,
            height: air_ctx.common_main.height() as u64,
            public_values: pvs[i].as_ptr(),
            num_cached: cached[i].len() as u32,
            _pad: 0,
            cached: cached[i].as_ptr(),
        })
        .collect();
    AbiTraces { traces, _public_values: pvs, _cached: cached }
}

fn key_of(mpk: &DeviceMultiStarkProvingKey<HipBackend>) -> *const ffi::OvmPk {
    mpk.per_air.first().expect("proving key without AIRs").other_data.key.0
}

impl TraceCommitter<HipBackend> for HipDevice {
    type Error = CommitError;

    /// hal.rs:84-87 -> `ovm_hip_commit_matrices`: matrices in, (root, PCS data) out; no key involved.
    fn commit(&self, traces: &[&HipMatrix<F>]) -> Result<(Digest, StackedPcsDataHip), Self::Error> {
        if traces.windows(2).any(|w| w[0].height() < w[1].height()) {
            return Err(CommitError::NotSorted);
        }
        let p = &self.params;
        let params = ffi::OvmPcsParams {
            l_skip: p.l_skip as u32,
            n_stack: p.n_stack as u32,
            log_blowup: p.log_blowup as u32,
            k_whir: p.k_whir() as u32,
        };
        let mats: Vec<ffi::OvmMatrix> = traces
            .iter()
            .map(|m| ffi::OvmMatrix { d_ptr: m.as_ptr() as *const u32, width: m.width() as u64, height: m.height() as u64 })
            .collect();
        let mut root = [0u32; 8];
        let mut handle: *mut ffi::OvmPcs = ptr::null_mut();
        // SAFETY: all pointers are valid for the call; the matrices are kept alive by the returned PCS data.
        HipError::from_result("ovm_hip_commit_matrices", unsafe {
            ffi::ovm_hip_commit_matrices(&params, mats.len() as u32, mats.as_ptr(), root.as_mut_ptr(), &mut handle, self.stream())
        })?;
        let commitment: [F; 8] = unsafe { std::mem::transmute(root) };
        let keep = traces.iter().map(|m| m.buffer().clone()).collect();
        Ok((commitment, StackedPcsDataHip::new(handle, commitment, keep)))
    }
}

impl MultiRapProver<HipBackend, DuplexSpongeHip> for HipDevice {
    type PartialProof = (GkrProof<SC>, BatchConstraintProof<SC>);
    type Artifacts = Vec<EF>;
    type Error = RapConstraintsError;

    /// hal.rs:94-112 -> `ovm_hip_prove_rap_constraints`: LogUp PoW + GKR, batch sumcheck, column
    /// openings. The transcript's sponge state goes in and comes back; the two proof parts return in
    /// the reference's own byte encoding (proof.rs:259-305) and are read with its `Decode`.
    fn prove_rap_constraints(
        &self,
        transcript: &mut DuplexSpongeHip,
        mpk: &DeviceMultiStarkProvingKey<HipBackend>,
        ctx: &ProvingContext<HipBackend>,
        _common_main_pcs_data: &StackedPcsDataHip,
    ) -> Result<(Self::PartialProof, Vec<EF>), Self::Error> {
        let abi = abi_traces(ctx);
        let mut ends = [0u64; 5];
        // SAFETY: valid key and trace list; host only.
        HipError::from_result("ovm_hip_proof_size", unsafe {
            ffi::ovm_hip_proof_size(key_of(mpk), abi.traces.len() as u32, abi.traces.as_ptr(), ends.as_mut_ptr())
        })?;
        let mut out = vec![0u8; (ends[2] - ends[0]) as usize];
        let (mut gkr_len, mut total_len, mut r_len) = (0u64, 0u64, 0u32);
        let r_cap = 1 + self.params.n_stack;
        let mut r_words = vec![0u32; 4 * r_cap];
        HipError::from_result("ovm_hip_prove_rap_constraints", unsafe {
            ffi::ovm_hip_prove_rap_constraints(
                key_of(mpk),
                abi.traces.len() as u32,
                abi.traces.as_ptr(),
                transcript.state_mut(),
                out.as_mut_ptr(),
                out.len() as u64,
                &mut gkr_len,
                &mut total_len,
                r_words.as_mut_ptr(),
                r_cap as u32,
                &mut r_len,
                self.stream(),
            )
        })?;
        let gkr = GkrProof::<SC>::decode_from_bytes(&out[..gkr_len as usize])?;
        let batch = BatchConstraintProof::<SC>::decode_from_bytes(&out[gkr_len as usize..total_len as usize])?;
        let r = r_words[..4 * r_len as usize]
            .chunks_exact(4)
            .map(|c| unsafe { std::mem::transmute::<[u32; 4], EF>([c[0], c[1], c[2], c[3]]) })
            .collect();
        Ok(((gkr, batch), r))
    }
}

impl OpeningProver<HipBackend, DuplexSpongeHip> for HipDevice {
    type OpeningProof = (StackingProof<SC>, WhirProof<SC>);
    type OpeningPoints = Vec<EF>;
    type Error = OpeningError;

    /// hal.rs:118-138 -> `ovm_hip_prove_openings`: stacked opening reduction + WHIR over the common
    /// main, the key's preprocessed commitments and the traces' cached mains. Consumes `ctx` and the
    /// common-main PCS data, as the trait says (they are dropped when the call has returned).
    fn prove_openings(
        &self,
        transcript: &mut DuplexSpongeHip,
        mpk: &DeviceMultiStarkProvingKey<HipBackend>,
        ctx: ProvingContext<HipBackend>,
        common_main_pcs_data: StackedPcsDataHip,
        r: Vec<EF>,
    ) -> Result<Self::OpeningProof, Self::Error> {
        let abi = abi_traces(&ctx);
        let mut ends = [0u64; 5];
        HipError::from_result("ovm_hip_proof_size", unsafe {
            ffi::ovm_hip_proof_size(key_of(mpk), abi.traces.len() as u32, abi.traces.as_ptr(), ends.as_mut_ptr())
        })?;
        let mut out = vec![0u8; (ends[4] - ends[2]) as usize];
        let (mut stacking_len, mut total_len) = (0u64, 0u64);
        let r_words: Vec<u32> = r.iter().flat_map(|e| unsafe { std::mem::transmute::<EF, [u32; 4]>(*e) }).collect();
        HipError::from_result("ovm_hip_prove_openings", unsafe {
            ffi::ovm_hip_prove_openings(
                key_of(mpk),
                abi.traces.len() as u32,
                abi.traces.as_ptr(),
                common_main_pcs_data.handle,
                r_words.as_ptr(),
                r.len() as u32,
                transcript.state_mut(),
                out.as_mut_ptr(),
                out.len() as u64,
                &mut stacking_len,
                &mut total_len,
                self.stream(),
            )
        })?;
        let stacking = StackingProof::<SC>::decode_from_bytes(&out[..stacking_len as usize])?;
        let whir = WhirProof::<SC>::decode_from_bytes(&out[stacking_len as usize..total_len as usize])?;
        drop((ctx, common_main_pcs_data));
        Ok((stacking, whir))
    }
}

impl ProverDevice<HipBackend, DuplexSpongeHip> for HipDevice {
    type Error = ProverError;
    type DeviceCtx = HipDeviceCtx;
    fn device_ctx(&self) -> &HipDeviceCtx {
        &self.ctx
    }
}
