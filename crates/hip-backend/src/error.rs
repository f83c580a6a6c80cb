//! Errors of the HIP backend. Peer of crates/cuda-backend/src/error.rs:11-33 (per-stage `thiserror`
//! enums folding into one `ProverError`) and crates/cuda-common/src/error.rs:54-60 (`from_result`).
use std::ffi::CStr;

use thiserror::Error;

use crate::ffi;

/// A non-zero return code of the C ABI (a `hipError_t` value) with the library's message.
#[derive(Error, Debug, Clone, PartialEq, Eq)]
#[error("HIP backend call `{call}` failed: {code} ({name}) {message}")]
pub struct HipError {
    pub call: &'static str,
    pub code: i32,
    pub name: String,
    pub message: String,
}

impl HipError {
    /// `CudaError::from_result` of the CUDA peer: 0 is success.
    pub fn from_result(call: &'static str, code: i32) -> Result<(), Self> {
        if code == 0 {
            return Ok(());
        }
        // SAFETY: both functions return pointers to NUL-terminated strings owned by the library
        // (thread-local for the message) that stay valid until the next call on this thread.
        let (name, message) = unsafe {
            let n = ffi::ovm_hip_error_name(code);
            let m = ffi::ovm_hip_last_error();
            (
                if n.is_null() { String::new() } else { CStr::from_ptr(n).to_string_lossy().into_owned() },
                if m.is_null() { String::new() } else { CStr::from_ptr(m).to_string_lossy().into_owned() },
            )
        };
        Err(Self { call, code, name, message })
    }
}

#[derive(Error, Debug)]
pub enum CommitError {
    #[error(transparent)]
    Hip(#[from] HipError),
    #[error("traces passed to commit are not sorted by height (descending)")]
    NotSorted,
}

#[derive(Error, Debug)]
pub enum RapConstraintsError {
    #[error(transparent)]
    Hip(#[from] HipError),
    #[error("decoding the GKR / batch-constraint proof bytes failed: {0}")]
    Decode(#[from] std::io::Error),
}

#[derive(Error, Debug)]
pub enum OpeningError {
    #[error(transparent)]
    Hip(#[from] HipError),
    #[error("decoding the stacking / WHIR proof bytes failed: {0}")]
    Decode(#[from] std::io::Error),
}

/// `ProverDevice::Error` (hal.rs:65-80): every stage error converts into it.
#[derive(Error, Debug)]
pub enum ProverError {
    #[error("commit: {0}")]
    Commit(#[from] CommitError),
    #[error("rap constraints: {0}")]
    RapConstraints(#[from] RapConstraintsError),
    #[error("openings: {0}")]
    Openings(#[from] OpeningError),
    #[error(transparent)]
    Hip(#[from] HipError),
}
