//! Device-resident types of the backend: buffers, matrices, PCS data. Peer of
//! crates/cuda-backend/src/base.rs:8-12 (`DeviceMatrix`: column-major, row index fastest) and
//! crates/cuda-common/src/d_buffer.rs. Field elements are Montgomery `u32`s on both sides of the
//! boundary (`BabyBear` is `repr(transparent)` over its Montgomery word), so nothing is converted.
use std::{ffi::c_void, marker::PhantomData, sync::Arc};

use openvm_stark_backend::prover::MatrixDimensions;

use crate::{error::HipError, ffi, prelude::F};

/// Owned device allocation of `len` elements of `T`.
pub struct DeviceBuffer<T> {
    ptr: *mut T,
    len: usize,
    /// `Some`: the memory belongs to the device key (a preprocessed trace) and is kept alive by it, not freed here
    borrowed_from: Option<Arc<PkHandle>>,
}
// SAFETY: the allocation is plain device memory; access is ordered by the streams of the calls using it.
unsafe impl<T: Send> Send for DeviceBuffer<T> {}
unsafe impl<T: Sync> Sync for DeviceBuffer<T> {}

impl<T> DeviceBuffer<T> {
    pub fn with_capacity(len: usize) -> Result<Self, HipError> {
        let mut p: *mut c_void = std::ptr::null_mut();
        // SAFETY: `p` is a valid out-pointer.
        HipError::from_result("ovm_hip_malloc", unsafe {
            ffi::ovm_hip_malloc(&mut p, len.max(1) * std::mem::size_of::<T>())
        })?;
        Ok(Self { ptr: p as *mut T, len, borrowed_from: None })
    }
    /// A view of `len` elements that live inside the device key `owner` (never freed by this buffer).
    pub(crate) fn borrowed(ptr: *const T, len: usize, owner: Arc<PkHandle>) -> Self {
        Self { ptr: ptr as *mut T, len, borrowed_from: Some(owner) }
    }
    pub fn from_host(host: &[T], stream: ffi::OvmStream) -> Result<Self, HipError> {
        let buf = Self::with_capacity(host.len())?;
        // SAFETY: both ranges are `host.len() * size_of::<T>()` bytes long.
        HipError::from_result("ovm_hip_memcpy_h2d", unsafe {
            ffi::ovm_hip_memcpy_h2d(buf.ptr as *mut c_void, host.as_ptr() as *const c_void, std::mem::size_of_val(host), stream)
        })?;
        Ok(buf)
    }
    pub fn to_host(&self, stream: ffi::OvmStream) -> Result<Vec<T>, HipError>
    where
        T: Default + Clone,
    {
        let mut out = vec![T::default(); self.len];
        // SAFETY: as above; the copy synchronises the stream before returning host data.
        HipError::from_result("ovm_hip_memcpy_d2h", unsafe {
            ffi::ovm_hip_memcpy_d2h(out.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * std::mem::size_of::<T>(), stream)
        })?;
        HipError::from_result("ovm_hip_stream_synchronize", unsafe { ffi::ovm_hip_stream_synchronize(stream) })?;
        Ok(out)
    }
    pub fn as_ptr(&self) -> *const T {
        self.ptr
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
}
impl<T> Drop for DeviceBuffer<T> {
    fn drop(&mut self) {
        if self.borrowed_from.is_none() {
            // SAFETY: allocated by ovm_hip_malloc, freed once.
            unsafe { ffi::ovm_hip_free(self.ptr as *mut c_void) };
        }
    }
}

/// `PB::Matrix`: a column-major matrix on the device (`hal.rs:152`: the transporter receives
/// `ColMajorMatrix`, so the layout crosses unchanged). Shared so that a cached trace and its
/// PCS data can point at the same buffer.
#[derive(Clone)]
pub struct HipMatrix<T> {
    buffer: Arc<DeviceBuffer<T>>,
    height: usize,
    width: usize,
}
impl<T> HipMatrix<T> {
    pub fn new(buffer: Arc<DeviceBuffer<T>>, height: usize, width: usize) -> Self {
        assert_eq!(buffer.len(), height * width);
        Self { buffer, height, width }
    }
    pub fn buffer(&self) -> &Arc<DeviceBuffer<T>> {
        &self.buffer
    }
    pub fn as_ptr(&self) -> *const T {
        self.buffer.as_ptr()
    }
}
impl<T> MatrixDimensions for HipMatrix<T> {
    fn height(&self) -> usize {
        self.height
    }
    fn width(&self) -> usize {
        self.width
    }
}

/// `PB::PcsData`: the library's `ovm_pcs_t` (stacked matrix, codeword, Merkle layers) plus the
/// matrices it may alias (the library reads a trace in place when it fills the stacked matrix).
pub struct StackedPcsDataHip {
    pub(crate) handle: *mut ffi::OvmPcs,
    pub commitment: [F; 8],
    _keep_alive: Vec<Arc<DeviceBuffer<F>>>,
    /// `Some`: the handle is the key's own PCS data of a preprocessed trace -- borrowed, not freed here
    borrowed_from: Option<Arc<PkHandle>>,
    _marker: PhantomData<F>,
}
// SAFETY: the handle is only read by prover calls; the library's PCS data is immutable after commit.
unsafe impl Send for StackedPcsDataHip {}
unsafe impl Sync for StackedPcsDataHip {}

impl StackedPcsDataHip {
    pub(crate) fn new(handle: *mut ffi::OvmPcs, commitment: [F; 8], keep_alive: Vec<Arc<DeviceBuffer<F>>>) -> Self {
        Self { handle, commitment, _keep_alive: keep_alive, borrowed_from: None, _marker: PhantomData }
    }
    pub(crate) fn borrowed(handle: *mut ffi::OvmPcs, commitment: [F; 8], owner: Arc<PkHandle>) -> Self {
        Self { handle, commitment, _keep_alive: Vec::new(), borrowed_from: Some(owner), _marker: PhantomData }
    }
}
impl Drop for StackedPcsDataHip {
    fn drop(&mut self) {
        if self.borrowed_from.is_none() {
            // SAFETY: handle came from ovm_hip_commit_matrices, freed once.
            unsafe { ffi::ovm_hip_pcs_free(self.handle) };
        }
    }
}

/// The library's device-side proving key (`ovm_pk_t`): parsed DAGs, compiled register programs,
/// committed preprocessed traces. One per `MultiStarkProvingKey`, shared by all its AIRs.
pub struct PkHandle(pub(crate) *mut ffi::OvmPk);
// SAFETY: the key is immutable after `ovm_hip_pk_load` (the transporter synchronises the stream
// before handing it out, as data_transporter.rs:69-70 of the CUDA peer does) and the library
// supports one prover thread per stream on a shared key.
unsafe impl Send for PkHandle {}
unsafe impl Sync for PkHandle {}
impl Drop for PkHandle {
    fn drop(&mut self) {
        // SAFETY: handle came from ovm_hip_pk_load, freed once.
        unsafe { ffi::ovm_hip_pk_free(self.0) };
    }
}

/// `PB::OtherAirData`: every AIR of a key points at the key's one device handle (the stage entries of
/// the C ABI take the whole key plus AIR indices).
#[derive(Clone)]
pub struct AirDataHip {
    pub(crate) key: Arc<PkHandle>,
    pub(crate) air_idx: u32,
}
