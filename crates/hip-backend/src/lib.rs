//! `openvm-hip-backend`: the MI355X (gfx950) prover backend of the SWIRL proof system, a peer of
//! `openvm-cpu-backend` and `openvm-cuda-backend` behind the same `ProverBackend` / `ProverDevice`
//! traits (crates/stark-backend/src/prover/hal.rs:23-207). The Rust side is deliberately thin:
//! every prover stage is ONE call into `libopenvm_hip.so` (hand-written HIP, C ABI in
//! include/openvm_hip.h); proof parts come back in the reference's own byte encoding and are read
//! with its `Decode` impls, so `Coordinator::prove`, the transcript observations between the
//! stages and the proof type stay the reference's code.
//!
//! ```ignore
//! use openvm_hip_backend::HipEngine;
//! openvm_backend_tests::backend_test_suite!(HipEngine);   // tests/integration.rs
//! ```
// hipStream_t and the library handles are opaque pointers forwarded to FFI, never dereferenced in Rust.
#![allow(clippy::not_unsafe_ptr_arg_deref)]

pub mod base;
pub mod data_transporter;
pub mod ffi;
pub mod pk_blob;
pub mod shard;
pub mod transcript;

mod device;
mod engine;
mod error;

pub use base::{AirDataHip, DeviceBuffer, HipMatrix, StackedPcsDataHip};
pub use device::{HipBackend, HipDevice, HipDeviceCtx, HipProverConfig};
pub use engine::HipEngine;
pub use error::*;
pub use transcript::DuplexSpongeHip;

pub mod prelude {
    pub use openvm_stark_sdk::config::baby_bear_poseidon2::{BabyBearPoseidon2Config, Digest, D_EF, EF, F};
    pub type SC = BabyBearPoseidon2Config;
}
