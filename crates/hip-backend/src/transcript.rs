//! `DuplexSpongeHip`: the Fiat-Shamir transcript of the HIP engine. Host-side Poseidon2 duplex sponge
//! with a `repr(C)` state the device stages read and write back (`ovm_sponge_t`), and a proof-of-work
//! that searches on the GPU. Peer of crates/cuda-backend/src/sponge.rs:39-48,193 (`DeviceSpongeState`,
//! `DuplexSpongeGpu`); semantics of crates/stark-backend/src/transcript/duplex_sponge.rs:62-105.
use openvm_stark_backend::FiatShamirTranscript;
use openvm_stark_sdk::config::baby_bear_poseidon2::poseidon2_perm;
use p3_field::{PrimeCharacteristicRing, PrimeField32};
use p3_symmetric::Permutation;

use crate::{
    error::HipError,
    ffi,
    prelude::{Digest, F, SC},
};

const WIDTH: usize = 16;
const CHUNK: usize = 8;

#[derive(Clone, Debug, Default)]
pub struct DuplexSpongeHip {
    /// Montgomery words, exactly the library's `ovm_sponge_t`
    state: ffi::OvmSponge,
    stream: StreamPtr,
}
#[derive(Clone, Copy, Debug)]
struct StreamPtr(ffi::OvmStream);
impl Default for StreamPtr {
    fn default() -> Self {
        Self(std::ptr::null_mut())
    }
}
// SAFETY: a HIP stream handle may be used from any thread.
unsafe impl Send for StreamPtr {}
unsafe impl Sync for StreamPtr {}

impl DuplexSpongeHip {
    pub fn on_stream(stream: ffi::OvmStream) -> Self {
        Self { state: ffi::OvmSponge::default(), stream: StreamPtr(stream) }
    }
    pub(crate) fn state_mut(&mut self) -> *mut ffi::OvmSponge {
        &mut self.state
    }
    fn words(&mut self) -> &mut [F; WIDTH] {
        // SAFETY: `BabyBear` is `repr(transparent)` over its Montgomery `u32`.
        unsafe { &mut *(self.state.state.as_mut_ptr() as *mut [F; WIDTH]) }
    }
    fn permute(&mut self) {
        let perm = poseidon2_perm();
        perm.permute_mut(self.words());
    }
}

impl FiatShamirTranscript<SC> for DuplexSpongeHip {
    fn observe(&mut self, value: F) {
        let i = self.state.absorb_idx as usize;
        self.words()[i] = value;
        self.state.absorb_idx += 1;
        if self.state.absorb_idx as usize == CHUNK {
            self.permute();
            self.state.absorb_idx = 0;
            self.state.sample_idx = CHUNK as u32;
        }
    }

    fn sample(&mut self) -> F {
        if self.state.absorb_idx != 0 || self.state.sample_idx == 0 {
            self.permute();
            self.state.absorb_idx = 0;
            self.state.sample_idx = CHUNK as u32;
        }
        self.state.sample_idx -= 1;
        let i = self.state.sample_idx as usize;
        self.words()[i]
    }

    fn observe_commit(&mut self, digest: Digest) {
        for x in digest {
            self.observe(x);
        }
    }

    /// traits.rs:74-89 searches with `find_any` (any valid witness); the device returns the MINIMUM
    /// one, which makes proofs reproducible. A CPU run is comparable when it is made deterministic
    /// the same way (RAYON_NUM_THREADS=1, tools/gen_golden.rs).
    fn grind(&mut self, bits: usize) -> F {
        if bits == 0 {
            return F::ZERO;
        }
        let mut w = 0u32;
        // SAFETY: valid sponge state and out-pointer; the call synchronises the stream.
        HipError::from_result("ovm_hip_sponge_grind", unsafe {
            ffi::ovm_hip_sponge_grind(&self.state, bits as u32, &mut w, self.stream.0)
        })
        .expect("proof-of-work search failed");
        let witness = F::from_u32(w);
        assert!(self.check_witness(bits, witness));
        let _ = witness.as_canonical_u32();
        witness
    }
}
