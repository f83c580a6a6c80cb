//! Host -> device transport of keys, matrices and committed traces (hal.rs:141-207). Peer of
//! crates/cuda-backend/src/data_transporter.rs:35-96. Matrices cross as they are (`ColMajorMatrix`
//! of Montgomery words); a proving key crosses as the flat blob of `pk_blob.rs`, which the library
//! parses, compiles into register programs and -- for preprocessed traces -- commits.
use std::{ptr, sync::Arc};

use openvm_stark_backend::{
    keygen::types::MultiStarkProvingKey,
    prover::{
        stacked_pcs::StackedPcsData, ColMajorMatrix, CommittedTraceData, DeviceDataTransporter, DeviceMultiStarkProvingKey,
        DeviceStarkProvingKey, MatrixDimensions, TraceCommitter,
    },
};

use crate::{
    base::{AirDataHip, DeviceBuffer, HipMatrix, PkHandle, StackedPcsDataHip},
    error::HipError,
    ffi,
    pk_blob::pk_blob,
    prelude::{Digest, F, SC},
    HipBackend, HipDevice,
};

impl DeviceDataTransporter<SC, HipBackend> for HipDevice {
    fn transport_pk_to_device(&self, mpk: &MultiStarkProvingKey<SC>) -> DeviceMultiStarkProvingKey<HipBackend> {
        let blob = pk_blob(mpk);
        let mut handle: *mut ffi::OvmPk = ptr::null_mut();
        // SAFETY: the blob outlives the call; the library copies what it keeps. Commits the preprocessed
        // traces on this device's stream and synchronises (the key may be shared between threads).
        HipError::from_result("ovm_hip_pk_load", unsafe {
            ffi::ovm_hip_pk_load(blob.as_ptr(), blob.len() as u64, &mut handle, self.stream())
        })
        .expect("failed to load the proving key on the device");
        let key = Arc::new(PkHandle(handle));
        let per_air = mpk
            .per_air
            .iter()
            .enumerate()
            .map(|(air_idx, pk)| {
                // The preprocessed trace and its PCS data live inside the key handle (ovm_hip_pk_load uploaded and
                // committed them): the trait's (commitment, trace, data) triple is handed out as BORROWED views of
                // exactly those -- no second upload, no second commit (the CUDA peer unstacks the host data once,
                // data_transporter.rs:44-51).
                let preprocessed_data = pk.preprocessed_data.as_ref().map(|_| {
                    let (mut d_trace, mut width, mut height) = (ptr::null::<u32>(), 0u64, 0u64);
                    let mut root = [0u32; 8];
                    let mut pcs: *mut ffi::OvmPcs = ptr::null_mut();
                    HipError::from_result("ovm_hip_pk_preprocessed_data", unsafe {
                        ffi::ovm_hip_pk_preprocessed_data(key.0, air_idx as u32, &mut d_trace, &mut width, &mut height, root.as_mut_ptr(), &mut pcs)
                    })
                    .expect("preprocessed data missing in the key");
                    // SAFETY: `F` is `repr(transparent)` over its Montgomery word; `Digest` is `[F; 8]`.
                    let commitment: Digest = unsafe { std::mem::transmute::<[u32; 8], Digest>(root) };
                    let buffer = DeviceBuffer::borrowed(d_trace as *const F, (width * height) as usize, key.clone());
                    let trace = HipMatrix::new(Arc::new(buffer), height as usize, width as usize);
                    let data = StackedPcsDataHip::borrowed(pcs, commitment, key.clone());
                    CommittedTraceData { commitment, trace, data: Arc::new(data) }
                });
                DeviceStarkProvingKey {
                    air_name: pk.air_name.clone(),
                    vk: pk.vk.clone(),
                    preprocessed_data,
                    other_data: AirDataHip { key: key.clone(), air_idx: air_idx as u32 },
                }
            })
            .collect();
        DeviceMultiStarkProvingKey::new(
            per_air,
            mpk.trace_height_constraints.clone(),
            mpk.max_constraint_degree,
            mpk.params.clone(),
            mpk.vk_pre_hash,
        )
    }

    fn transport_matrix_to_device(&self, matrix: &ColMajorMatrix<F>) -> HipMatrix<F> {
        let buffer = DeviceBuffer::from_host(&matrix.values, self.stream()).expect("host -> device copy failed");
        self.device_ctx_sync();
        HipMatrix::new(Arc::new(buffer), matrix.height(), matrix.width())
    }

    /// The reference hands over host PCS data (stacked matrix + Merkle tree) computed by the CPU
    /// backend; the device form is rebuilt by committing the unstacked matrices again -- the
    /// commitment is a function of the data, so it is the same one.
    fn transport_pcs_data_to_device(&self, pcs_data: &StackedPcsData<F, Digest>) -> StackedPcsDataHip {
        let mats: Vec<HipMatrix<F>> = (0..pcs_data.layout.mat_starts.len())
            .map(|i| self.transport_matrix_to_device(&pcs_data.mat_view(i).to_matrix()))
            .collect();
        let refs: Vec<&HipMatrix<F>> = mats.iter().collect();
        let (commitment, data) = self.commit(&refs).expect("commit failed");
        debug_assert_eq!(commitment, pcs_data.commit().expect("host PCS data without a root"));
        data
    }

    fn transport_matrix_from_device_to_host(&self, matrix: &HipMatrix<F>) -> ColMajorMatrix<F> {
        let values = matrix.buffer().to_host(self.stream()).expect("device -> host copy failed");
        ColMajorMatrix::new(values, matrix.width())
    }
}

impl HipDevice {
    fn device_ctx_sync(&self) {
        // SAFETY: valid stream.
        HipError::from_result("ovm_hip_stream_synchronize", unsafe { ffi::ovm_hip_stream_synchronize(self.stream()) })
            .expect("stream synchronisation failed");
    }
}
