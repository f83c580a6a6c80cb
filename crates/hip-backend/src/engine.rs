//! `HipEngine`: the `StarkEngine` the SDK, the benches and `backend_test_suite!` select.
//! Peer of crates/cuda-backend/src/engine.rs:48-83.
use openvm_stark_backend::{prover::Coordinator, StarkEngine, SystemParams};
use openvm_stark_sdk::config::baby_bear_poseidon2::BabyBearPoseidon2Config;

use crate::{prelude::SC, DuplexSpongeHip, HipBackend, HipDevice};

pub struct HipEngine {
    device: HipDevice,
    config: BabyBearPoseidon2Config,
}

impl StarkEngine for HipEngine {
    type SC = SC;
    type PB = HipBackend;
    type PD = HipDevice;
    type TS = DuplexSpongeHip;

    fn new(params: SystemParams) -> Self {
        let config = BabyBearPoseidon2Config::default_from_params(params.clone());
        Self { device: HipDevice::new(params).expect("failed to create HipDevice"), config }
    }

    fn config(&self) -> &Self::SC {
        &self.config
    }

    fn device(&self) -> &Self::PD {
        &self.device
    }

    fn initial_transcript(&self) -> Self::TS {
        DuplexSpongeHip::on_stream(self.device.stream())
    }

    fn prover_from_transcript(&self, transcript: DuplexSpongeHip) -> Coordinator<SC, Self::PB, Self::PD, Self::TS> {
        Coordinator::new(HipBackend, self.device.clone(), transcript)
    }
}
