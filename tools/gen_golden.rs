//! gen_golden — dump reference proofs for the fixtures of tests/airs.py.
//!
//! STATUS: NOT COMPILED. This image has no cargo/rustc; the file is written against
//! openvm-org/stark-backend v2.0.1 by reading its sources and has never been built. Expect to fix
//! import paths when first compiling it.
//!
//! Build (on a machine with cargo, inside a checkout of the reference):
//!     cp tools/gen_golden.rs crates/stark-sdk/examples/gen_golden.rs
//!     RAYON_NUM_THREADS=1 cargo run --release -p openvm-stark-sdk --example gen_golden -- OUT_DIR
//! RAYON_NUM_THREADS=1 makes `grind`'s `find_any` return the smallest witness
//! (crates/stark-backend/src/transcript/traits.rs:74-89), which is what this backend returns.
//!
//! Output: one JSON file per fixture in OUT_DIR with
//!   "pk_blob":  the flat u32 proving-key blob documented in include/openvm_hip.h
//!               (`ovm_hip_pk_load`), field constants canonical;
//!   "traces":   [{"air_idx", "height", "values" (column-major, canonical), "public_values"}];
//!   "proof_hex": hex of `Proof::encode_to_vec()` (crates/stark-backend/src/codec.rs:29);
//! plus "poseidon2_kat.json": permutation of [0, 1, ..., 15] (canonical in / out).
//! Copy the files to tests/golden/reference/; tests/test_oracle_golden.py then requires the oracle
//! (and through tests/test_gpu_prove.py the HIP prover) to reproduce every proof byte for byte.
use std::{fs, path::PathBuf};

use openvm_stark_backend::{
    air_builders::symbolic::{
        symbolic_variable::Entry, SymbolicExpressionNode as Node,
    },
    codec::Encode,
    keygen::types::MultiStarkVerifyingKey,
    p3_field::{PrimeCharacteristicRing, PrimeField32},
    prover::{ColMajorMatrix, MatrixDimensions, ProvingContext},
    test_utils::{default_test_params_small, FibFixture, InteractionsFixture11, TestFixture},
    StarkEngine, SystemParams,
};
use openvm_stark_sdk::config::baby_bear_poseidon2::{
    default_babybear_poseidon2_16, BabyBearPoseidon2Config as SC, BabyBearPoseidon2CpuEngine,
};
use p3_baby_bear::BabyBear;
use p3_symmetric::Permutation;

const MAGIC: u32 = 0x4F56_4D31;

fn canon(x: &BabyBear) -> u32 {
    x.as_canonical_u32()
}

/// The blob `ovm_hip_pk_load` parses (include/openvm_hip.h) — also what a real
/// `transport_pk_to_device` for the HIP backend would send.
fn pk_blob(vk: &MultiStarkVerifyingKey<SC>) -> Vec<u32> {
    let p: &SystemParams = &vk.inner.params;
    let mut w: Vec<u32> = vec![
        MAGIC,
        p.l_skip as u32,
        p.n_stack as u32,
        p.w_stack as u32,
        p.log_blowup as u32,
        p.whir.k as u32,
        p.whir.mu_pow_bits as u32,
        p.whir.query_phase_pow_bits as u32,
        p.whir.folding_pow_bits as u32,
        p.logup.pow_bits as u32,
        p.max_constraint_degree as u32,
        p.whir.rounds.len() as u32,
    ];
    w.extend(p.whir.rounds.iter().map(|r| r.num_queries as u32));
    w.extend(vk.pre_hash.iter().map(canon));
    w.push(vk.inner.per_air.len() as u32);
    for air in &vk.inner.per_air {
        let width = &air.params.width;
        assert!(width.preprocessed.is_none() && width.cached_mains.is_empty(),
                "preprocessed / cached traces are not supported by the HIP backend yet");
        w.extend([
            air.params.need_rot as u32,
            air.is_required as u32,
            air.max_constraint_degree as u32,
            air.params.num_public_values as u32,
            0, // has_preprocessed
            0, // preprocessed_width
            0, // num_cached
            width.common_main as u32,
        ]);
        let dag = &air.symbolic_constraints.constraints;
        w.push(dag.nodes.len() as u32);
        for n in &dag.nodes {
            // (kind, entry, part_index, offset, x, y) — kinds as in include/openvm_hip.h
            let rec: [u32; 6] = match n {
                Node::Variable(v) => {
                    let (entry, part, off) = match v.entry {
                        Entry::Preprocessed { offset } => (0, 0, offset),
                        Entry::Main { part_index, offset } => (1, part_index, offset),
                        Entry::Public => (2, 0, 0),
                        Entry::Challenge => panic!("challenge variables do not occur in v2 AIRs"),
                    };
                    [0, entry, part as u32, off as u32, v.index as u32, 0]
                }
                Node::IsFirstRow => [1, 0, 0, 0, 0, 0],
                Node::IsLastRow => [2, 0, 0, 0, 0, 0],
                Node::IsTransition => [3, 0, 0, 0, 0, 0],
                Node::Constant(c) => [4, 0, 0, 0, canon(c), 0],
                Node::Add { left_idx, right_idx, .. } => [5, 0, 0, 0, *left_idx as u32, *right_idx as u32],
                Node::Sub { left_idx, right_idx, .. } => [6, 0, 0, 0, *left_idx as u32, *right_idx as u32],
                Node::Neg { idx, .. } => [7, 0, 0, 0, *idx as u32, 0],
                Node::Mul { left_idx, right_idx, .. } => [8, 0, 0, 0, *left_idx as u32, *right_idx as u32],
            };
            w.extend(rec);
        }
        w.push(dag.constraint_idx.len() as u32);
        w.extend(dag.constraint_idx.iter().map(|&i| i as u32));
        let ints = &air.symbolic_constraints.interactions;
        w.push(ints.len() as u32);
        for it in ints {
            w.extend([it.bus_index as u32, it.count_weight as u32, it.count as u32, it.message.len() as u32]);
            w.extend(it.message.iter().map(|&m| m as u32));
        }
    }
    w
}

fn dump<Fx: TestFixture<SC>>(out: &PathBuf, name: &str, params: SystemParams, fx: Fx) {
    let engine = BabyBearPoseidon2CpuEngine::new(params);
    let (pk, vk) = fx.keygen(&engine);
    // the context the prover sees (column-major matrices), for the "traces" section
    let ctx: ProvingContext<_> = fx.generate_proving_ctx();
    let traces: Vec<serde_json::Value> = ctx
        .per_trace
        .iter()
        .map(|(air_idx, c)| {
            let m: &ColMajorMatrix<BabyBear> = &c.common_main;
            serde_json::json!({
                "air_idx": air_idx,
                "height": m.height(),
                "values": m.values.iter().map(canon).collect::<Vec<_>>(),
                "public_values": c.public_values.iter().map(canon).collect::<Vec<_>>(),
            })
        })
        .collect();
    let proof = fx.prove(&engine, &pk);
    engine.verify(&vk, &proof).expect("reference proof must verify");
    let bytes = proof.encode_to_vec().unwrap();
    let hex: String = bytes.iter().map(|b| format!("{b:02x}")).collect();
    let doc = serde_json::json!({
        "fixture": name,
        "reference": "openvm-org/stark-backend v2.0.1, BabyBearPoseidon2CpuEngine, RAYON_NUM_THREADS=1",
        "pk_blob": pk_blob(&vk),
        "traces": traces,
        "proof_hex": hex,
    });
    fs::write(out.join(format!("{name}.json")), serde_json::to_vec(&doc).unwrap()).unwrap();
    eprintln!("{name}: {} proof bytes", bytes.len());
}

fn main() {
    assert_eq!(std::env::var("RAYON_NUM_THREADS").as_deref(), Ok("1"),
               "run with RAYON_NUM_THREADS=1 so that PoW witnesses are the minimum ones");
    let out = PathBuf::from(std::env::args().nth(1).expect("usage: gen_golden OUT_DIR"));
    fs::create_dir_all(&out).unwrap();

    // Poseidon2 known-answer vector (closes the oracle's missing KAT)
    let perm = default_babybear_poseidon2_16();
    let mut st: [BabyBear; 16] = core::array::from_fn(|i| BabyBear::from_u32(i as u32));
    perm.permute_mut(&mut st);
    fs::write(out.join("poseidon2_kat.json"),
              serde_json::to_vec(&serde_json::json!({
                  "input": (0..16).collect::<Vec<u32>>(),
                  "output": st.iter().map(canon).collect::<Vec<_>>() })).unwrap()).unwrap();

    // Fixtures mirrored in tests/airs.py (crates/stark-backend/src/test_utils/mod.rs:152-330)
    dump(&out, "fib_2pow4_small", default_test_params_small(), FibFixture::new(0, 1, 1 << 4));
    dump(&out, "fib_2pow6_small", default_test_params_small(), FibFixture::new(0, 1, 1 << 6));
    dump(&out, "fib_2pow8_testing12", SystemParams::new_for_testing(12), FibFixture::new(0, 1, 1 << 8));
    dump(&out, "interactions11_small", default_test_params_small(), InteractionsFixture11);
}
