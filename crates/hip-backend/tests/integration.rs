//! The reference's backend conformance suite (crates/backend-tests/src/lib.rs:1502-1712) on the HIP
//! engine, exactly as crates/cpu-backend/tests/integration.rs:5 and crates/cuda-backend/src/tests.rs:78
//! run it on theirs. Needs an MI355X and `libopenvm_hip.so` (build.rs).
use openvm_hip_backend::HipEngine;

type Engine = HipEngine;

openvm_backend_tests::backend_test_suite!(Engine);

/// Bit-identity against `openvm-cpu-backend` on the fixtures of tools/gen_golden.rs: the CPU engine
/// run with one rayon thread (its proof-of-work search then returns the minimum witness, as the
/// device search always does) and the HIP engine must produce the same `Proof::encode_to_vec`.
#[cfg(test)]
mod bit_identity {
    use openvm_stark_backend::{codec::Encode, test_utils::{FibFixture, InteractionsFixture11, TestFixture}, StarkEngine};
    use openvm_stark_sdk::config::{app_params_with_100_bits_security, baby_bear_poseidon2::BabyBearPoseidon2CpuEngine};

    use super::Engine;

    fn same_bytes<Fx: TestFixture<openvm_hip_backend::prelude::SC>>(fx: Fx, log_stacked_height: usize) {
        let params = app_params_with_100_bits_security(log_stacked_height);
        let cpu = BabyBearPoseidon2CpuEngine::new(params.clone());
        let hip = Engine::new(params);
        let (_, cpu_proof) = fx.keygen_and_prove(&cpu);
        let (_, hip_proof) = fx.keygen_and_prove(&hip);
        assert_eq!(cpu_proof.encode_to_vec().unwrap(), hip_proof.encode_to_vec().unwrap());
    }

    #[test]
    fn fibonacci_2_12() {
        std::env::set_var("RAYON_NUM_THREADS", "1");
        same_bytes(FibFixture::new(0, 1, 1 << 12), 12);
    }

    #[test]
    fn interactions() {
        std::env::set_var("RAYON_NUM_THREADS", "1");
        same_bytes(InteractionsFixture11, 12);
    }
}
