//! gen_golden — dump REFERENCE proofs for the fixtures of tests/conformance.py and tests/airs.py.
//!
//! STATUS: NOT COMPILED. This image has no cargo/rustc; the file is written against openvm-org/stark-backend v2.0.1 by
//! reading its sources and has never been built. Expect to fix import paths when first compiling it.
//! `tools/run_on_cargo_box.sh <reference checkout>` copies it into place, runs it and installs its output.
//!
//! Build by hand (on a machine with cargo, inside a checkout of the reference):
//!     cp tools/gen_golden.rs crates/stark-sdk/examples/gen_golden.rs
//!     RAYON_NUM_THREADS=1 cargo run --release -p openvm-stark-sdk --example gen_golden -- OUT_DIR
//! RAYON_NUM_THREADS=1 makes `grind`'s `find_any` return the smallest witness
//! (crates/stark-backend/src/transcript/traits.rs:74-89), which is what this backend returns.
//!
//! Output: one JSON file per case in OUT_DIR with
//!   "pk_blob":   the flat u32 proving-key blob documented in include/openvm_hip.h (`ovm_hip_pk_load`; same function as
//!                crates/hip-backend/src/pk_blob.rs), field constants canonical, preprocessed traces in its PREP section;
//!   "traces":    [{"air_idx", "height", "values" (common main, column-major, canonical), "public_values",
//!                 "cached": [[values of cached main 0], ...]}] -- only the AIRs PRESENT in the context;
//!   "proof_hex": hex of `Proof::encode_to_vec()` (crates/stark-backend/src/codec.rs:29), or null with
//!   "prover_error": the Display of the prover's error for the negative cases;
//! plus "poseidon2_kat.json": permutation of [0, 1, ..., 15] (canonical in / out).
//! The case NAMES are those of tests/conformance.py::suite() -- `tests/test_oracle_golden.py` pairs them up, requires the
//! restated fixtures to produce the very same key blob and traces, and the oracle (and through tests/test_gpu_prove.py the
//! HIP prover) to reproduce every proof byte for byte.
use std::{fs, path::PathBuf, sync::Arc};

use itertools::Itertools;
use openvm_stark_backend::{
    air_builders::symbolic::{symbolic_variable::Entry, SymbolicExpressionNode as Node},
    codec::Encode,
    keygen::types::MultiStarkProvingKey,
    p3_field::{PrimeCharacteristicRing, PrimeField32},
    prover::{AirProvingContext, ColMajorMatrix, CpuColMajorBackend, MatrixDimensions, ProvingContext},
    test_utils::{
        default_test_params_small,
        dummy_airs::{
            fib_air::air::FibonacciAir,
            interaction::dummy_interaction_air::{DummyInteractionAir, DummyInteractionChip, DummyInteractionData},
        },
        test_system_params_small, CachedFixture11, FibFixture, InteractionsFixture11, MixtureFixture,
        PreprocessedAndCachedFixture, PreprocessedFibFixture, SelfInteractionFixture, TestFixture,
    },
    AirRef, StarkEngine, SystemParams, WhirConfig, WhirParams, WhirProximityStrategy,
};
use openvm_stark_sdk::config::{
    app_params_with_100_bits_security,
    baby_bear_poseidon2::{default_babybear_poseidon2_16, BabyBearPoseidon2Config as SC, BabyBearPoseidon2CpuEngine},
    log_up_params::log_up_security_params_baby_bear_100_bits,
};
use p3_baby_bear::BabyBear;
use p3_matrix::dense::RowMajorMatrix;
use p3_symmetric::Permutation;

type F = BabyBear;
type Engine = BabyBearPoseidon2CpuEngine;
type Ctx = ProvingContext<CpuColMajorBackend<SC>>;

const MAGIC: u32 = 0x4F56_4D31;
const PREP_MAGIC: u32 = 0x5052_4550;

fn canon(x: &F) -> u32 {
    x.as_canonical_u32()
}

/// The blob `ovm_hip_pk_load` parses (include/openvm_hip.h); the same words crates/hip-backend/src/pk_blob.rs sends.
fn pk_blob(mpk: &MultiStarkProvingKey<SC>) -> Vec<u32> {
    let p: &SystemParams = &mpk.params;
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
        mpk.max_constraint_degree as u32,
        p.whir.rounds.len() as u32,
    ];
    w.extend(p.whir.rounds.iter().map(|r| r.num_queries as u32));
    w.extend(mpk.vk_pre_hash.iter().map(canon));
    w.push(mpk.per_air.len() as u32);
    for air in &mpk.per_air {
        let vk = &air.vk;
        let width = &vk.params.width;
        w.extend([
            vk.params.need_rot as u32,
            vk.is_required as u32,
            vk.max_constraint_degree as u32,
            vk.params.num_public_values as u32,
            width.preprocessed.is_some() as u32,
            width.preprocessed.unwrap_or(0) as u32,
            width.cached_mains.len() as u32,
        ]);
        w.extend(width.cached_mains.iter().map(|&c| c as u32));
        w.push(width.common_main as u32);
        let dag = &vk.symbolic_constraints.constraints;
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
        let ints = &vk.symbolic_constraints.interactions;
        w.push(ints.len() as u32);
        for it in ints {
            w.extend([it.bus_index as u32, it.count_weight as u32, it.count as u32, it.message.len() as u32]);
            w.extend(it.message.iter().map(|&m| m as u32));
        }
    }
    if mpk.per_air.iter().any(|a| a.preprocessed_data.is_some()) {
        w.push(PREP_MAGIC);
        for air in &mpk.per_air {
            if let Some(pre) = &air.preprocessed_data {
                // the committed preprocessed trace, column-major (keygen/mod.rs:283-300)
                let m = pre.mat_view(0).to_matrix();
                w.push(m.height() as u32);
                w.extend(m.values.iter().map(canon));
            }
        }
    }
    w
}

fn traces_json(ctx: &Ctx) -> Vec<serde_json::Value> {
    ctx.per_trace
        .iter()
        .map(|(air_idx, c)| {
            let m: &ColMajorMatrix<F> = &c.common_main;
            serde_json::json!({
                "air_idx": air_idx,
                "height": m.height(),
                "values": m.values.iter().map(canon).collect::<Vec<_>>(),
                "public_values": c.public_values.iter().map(canon).collect::<Vec<_>>(),
                "cached": c.cached_mains.iter()
                    .map(|d| d.trace.values.iter().map(canon).collect::<Vec<_>>()).collect::<Vec<_>>(),
            })
        })
        .collect()
}

/// keygen with `StarkEngine::keygen` (every AIR optional), prove `ctx`, verify, write the case.
fn dump_ctx(out: &PathBuf, name: &str, params: SystemParams, airs: Vec<AirRef<SC>>, ctx: Ctx, expect_ok: bool) {
    let engine = Engine::new(params);
    let (pk, vk) = engine.keygen(&airs);
    let traces = traces_json(&ctx);
    let device = engine.device();
    let d_pk = device.transport_pk_to_device(&pk);
    let d_ctx = device.transport_proving_ctx_to_device(&ctx);
    let (proof_hex, err) = match engine.prove(&d_pk, d_ctx) {
        Ok(proof) => {
            assert!(expect_ok, "{name}: the reference prover accepted a negative case");
            engine.verify(&vk, &proof).expect("reference proof must verify");
            let bytes = proof.encode_to_vec().unwrap();
            eprintln!("{name}: {} proof bytes", bytes.len());
            (Some(bytes.iter().map(|b| format!("{b:02x}")).collect::<String>()), None)
        }
        Err(e) => {
            assert!(!expect_ok, "{name}: the reference prover failed: {e}");
            eprintln!("{name}: prover error (expected): {e}");
            (None, Some(format!("{e}")))
        }
    };
    let doc = serde_json::json!({
        "fixture": name,
        "reference": "openvm-org/stark-backend v2.0.1, BabyBearPoseidon2CpuEngine, RAYON_NUM_THREADS=1",
        "pk_blob": pk_blob(&pk),
        "traces": traces,
        "proof_hex": proof_hex,
        "prover_error": err,
    });
    let file = name.replace(['[', ']'], "_");
    fs::write(out.join(format!("{file}.json")), serde_json::to_vec(&doc).unwrap()).unwrap();
}

fn dump<Fx: TestFixture<SC>>(out: &PathBuf, name: &str, params: SystemParams, fx: Fx) {
    dump_ctx(out, name, params, fx.airs(), fx.generate_proving_ctx(), true);
}

fn row_major_ctx(vals: &[usize], width: usize) -> AirProvingContext<CpuColMajorBackend<SC>> {
    let rm = RowMajorMatrix::new(vals.iter().map(|&v| F::from_usize(v)).collect(), width);
    AirProvingContext::simple_no_pis(ColMajorMatrix::from_row_major(&rm))
}

fn sender_receiver(out: &PathBuf, name: &str, senders: &[&[usize]], receivers: &[&[usize]], expect_ok: bool) {
    let mut airs: Vec<AirRef<SC>> = vec![];
    let mut per_trace = vec![];
    for s in senders {
        airs.push(Arc::new(DummyInteractionAir::new(1, true, 0)));
        per_trace.push(row_major_ctx(s, 2));
    }
    for r in receivers {
        airs.push(Arc::new(DummyInteractionAir::new(1, false, 0)));
        per_trace.push(row_major_ctx(r, 2));
    }
    let ctx = ProvingContext::new(per_trace.into_iter().enumerate().collect());
    dump_ctx(out, name, default_test_params_small(), airs, ctx, expect_ok);
}

/// fib_air_roundtrip's parameters (crates/backend-tests/src/lib.rs:262-282)
fn fib_roundtrip_params(l_skip: usize) -> SystemParams {
    let (n_stack, k_whir, log_blowup) = (8, 4, 1);
    let whir_params = WhirParams {
        k: k_whir,
        log_final_poly_len: k_whir,
        query_phase_pow_bits: 1,
        folding_pow_bits: 11,
        mu_pow_bits: 10,
        proximity: WhirProximityStrategy::UniqueDecoding,
    };
    SystemParams {
        l_skip,
        n_stack,
        w_stack: 8,
        log_blowup,
        whir: WhirConfig::new(log_blowup, l_skip + n_stack, whir_params, 80),
        logup: log_up_security_params_baby_bear_100_bits(0.0),
        max_constraint_degree: 3,
    }
}

fn chip(field_width: usize, is_send: bool, count: Vec<u32>, fields: Vec<Vec<u32>>, partition: Option<SC>) -> DummyInteractionChip<SC> {
    let mut c = match partition {
        Some(config) => DummyInteractionChip::new_with_partition(config, field_width, is_send, 0),
        None => DummyInteractionChip::new_without_partition(field_width, is_send, 0),
    };
    c.load_data(DummyInteractionData { count, fields });
    c
}

fn main() {
    assert_eq!(std::env::var("RAYON_NUM_THREADS").as_deref(), Ok("1"),
               "run with RAYON_NUM_THREADS=1 so that PoW witnesses are the minimum ones");
    let out = PathBuf::from(std::env::args().nth(1).expect("usage: gen_golden OUT_DIR"));
    fs::create_dir_all(&out).unwrap();
    let small = default_test_params_small;
    let config = || Engine::new(small()).config().clone();

    // Poseidon2 known-answer vector of the DEFAULT constants
    let perm = default_babybear_poseidon2_16();
    let mut st: [F; 16] = core::array::from_fn(|i| F::from_u32(i as u32));
    perm.permute_mut(&mut st);
    fs::write(out.join("poseidon2_kat.json"),
              serde_json::to_vec(&serde_json::json!({
                  "input": (0..16).collect::<Vec<u32>>(),
                  "output": st.iter().map(canon).collect::<Vec<_>>() })).unwrap()).unwrap();

    // ---- the cases of tests/conformance.py::suite(), same names (crates/backend-tests/src/lib.rs) ----
    dump(&out, "interactions_single_sender_receiver_happy", small(), InteractionsFixture11);
    dump(&out, "single_cached_trace_stark", small(), CachedFixture11::new(config()));
    for lg in [10usize, 3, 2, 1, 0] {
        let sels = (0..1usize << lg).map(|i| i % 2 == 0).collect_vec();
        dump(&out, &format!("single_preprocessed_trace_stark[{lg}]"), small(), PreprocessedFibFixture::new(0, 1, sels));
        dump(&out, &format!("multi_interaction_traces_stark[{lg}]"), small(),
             SelfInteractionFixture { widths: vec![4, 7, 8, 8, 10, 100], log_height: lg, bus_index: 4 });
        dump(&out, &format!("mixture_traces_stark[{lg}]"), small(), MixtureFixture::standard(lg, config()));
    }
    dump(&out, "matrix_stacking_overflow", test_system_params_small(3, 5, 3),
         SelfInteractionFixture { widths: vec![4, 7, 8, 8, 10], log_height: 1, bus_index: 4 });
    for (l_skip, lg) in [(2usize, 10usize), (2, 1), (2, 0), (3, 2), (6, 10)] {
        dump(&out, &format!("fib_air_roundtrip[{l_skip}-{lg}]"), fib_roundtrip_params(l_skip), FibFixture::new(0, 1, 1 << lg));
    }
    for (a, b, c) in [(2usize, 8usize, 3usize), (5, 5, 4)] {
        let p = || test_system_params_small(a, b, c);
        dump(&out, &format!("dummy_interactions_roundtrip[{a}-{b}-{c}]"), p(), InteractionsFixture11);
        let sels = (0..1usize << 8).map(|i| i % 2 == 0).collect_vec();
        dump(&out, &format!("preprocessed_trace_roundtrip[{a}-{b}-{c}]"), p(), PreprocessedFibFixture::new(0, 1, sels.clone()));
        for parts in 1..=3usize {
            let cfg = Engine::new(p()).config().clone();
            dump(&out, &format!("preprocessed_and_cached_trace_roundtrip[{a}-{b}-{c}-{parts}]"), p(),
                 PreprocessedAndCachedFixture::new(sels.clone(), cfg, parts));
        }
    }
    for (a, b, c) in [(2usize, 8usize, 3usize), (5, 5, 4), (5, 8, 3), (6, 7, 3)] {
        let p = test_system_params_small(a, b, c);
        let cfg = Engine::new(p.clone()).config().clone();
        dump(&out, &format!("cached_trace_roundtrip[{a}-{b}-{c}]"), p, CachedFixture11::new(cfg));
    }
    // single_fib_and_dummy_trace_stark (lib.rs:571-640)
    for lg in [3usize, 2, 1, 0] {
        let sender_height = 2 * (1 << 3);
        let cyc = |pat: &[usize], n: usize| pat.iter().copied().cycle().take(n).collect_vec();
        let s = cyc(&[0, 1, 3, 5, 7, 4, 546, 889], 2 * sender_height);
        let r = cyc(&[1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 889, 1, 889, 0, 456], 4 * sender_height);
        let fib = FibFixture::new(0, 1, 2 * (1 << lg));
        let mut airs = <InteractionsFixture11 as TestFixture<SC>>::airs(&InteractionsFixture11);
        airs.extend(<FibFixture as TestFixture<SC>>::airs(&fib));
        let fib_ctx = <FibFixture as TestFixture<SC>>::generate_proving_ctx(&fib).per_trace.pop().unwrap().1;
        let ctx = ProvingContext::new(vec![(0, row_major_ctx(&s, 2)), (1, row_major_ctx(&r, 2)), (2, fib_ctx)]);
        dump_ctx(&out, &format!("single_fib_and_dummy_trace_stark[{lg}]"), small(), airs, ctx, true);
    }
    // optional_air (lib.rs:642-750): ONE key, three contexts
    {
        let key = || -> Vec<AirRef<SC>> {
            vec![Arc::new(FibonacciAir),
                 chip(1, true, vec![1], vec![vec![1]], None).air(),
                 chip(1, true, vec![1], vec![vec![1]], Some(config())).air(),
                 chip(1, false, vec![1], vec![vec![1]], None).air()]
        };
        let f3 = || vec![vec![1u32], vec![2], vec![3]];
        let fib = <FibFixture as TestFixture<SC>>::generate_proving_ctx(&FibFixture::new(0, 1, 8)).per_trace.pop().unwrap().1;
        let c1 = ProvingContext::new(vec![
            (0, fib),
            (1, chip(1, true, vec![1, 2, 4], f3(), None).generate_proving_ctx()),
            (2, chip(1, true, vec![1, 2, 8], f3(), Some(config())).generate_proving_ctx()),
            (3, chip(1, false, vec![2, 4, 12], f3(), None).generate_proving_ctx()),
        ]);
        dump_ctx(&out, "optional_air[case1]", small(), key(), c1, true);
        let c2 = ProvingContext::new(vec![
            (1, chip(1, true, vec![1, 2, 4], f3(), None).generate_proving_ctx()),
            (3, chip(1, false, vec![1, 2, 4], f3(), None).generate_proving_ctx()),
        ]);
        dump_ctx(&out, "optional_air[case2]", small(), key(), c2, true);
        let c3 = ProvingContext::new(vec![(3, chip(1, false, vec![1, 2, 4], f3(), None).generate_proving_ctx())]);
        dump_ctx(&out, "optional_air[case3]", small(), key(), c3, false);
    }
    let recv: &[usize] = &[1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 889, 1, 889, 0, 456];
    sender_receiver(&out, "interaction_multi_rows_neg", &[&[0, 1, 3, 5, 7, 4, 546, 0]],
                    &[&[1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 0, 0, 0, 0, 456]], false);
    sender_receiver(&out, "interaction_all_zero_sender", &[&[0, 1, 0, 5, 0, 4, 0, 889]], &[], true);
    sender_receiver(&out, "interaction_multi_senders", &[&[0, 1, 3, 5, 6, 4, 333, 889], &[1, 4, 213, 889]], &[recv], true);
    sender_receiver(&out, "interaction_multi_senders_neg", &[&[0, 1, 3, 5, 5, 4, 333, 889], &[1, 4, 213, 889]], &[recv], false);
    sender_receiver(&out, "interaction_multi_sender_receiver", &[&[0, 1, 3, 5, 6, 4, 333, 889], &[1, 4, 213, 889]],
                    &[&[1, 5, 3, 4, 4, 4, 2, 5, 0, 123, 545, 889, 0, 289, 0, 456], &[1, 889]], true);
    // interaction_cached_trace_neg (lib.rs:1020-1074): AIR order [partitioned receiver, sender]
    {
        let s = chip(2, true, vec![0, 7, 3, 546], vec![vec![1, 1], vec![4, 2], vec![5, 1], vec![889, 4]], None);
        let r = chip(2, false, vec![1, 3, 4, 2, 0, 545, 1, 0],
                     vec![vec![5, 1], vec![4, 2], vec![4, 2], vec![5, 1], vec![123, 3], vec![889, 4], vec![889, 10], vec![456, 5]],
                     Some(config()));
        let ctx = ProvingContext::new(vec![(0, r.generate_proving_ctx()), (1, s.generate_proving_ctx())]);
        dump_ctx(&out, "interaction_cached_trace_neg", small(), vec![r.air(), s.air()], ctx, false);
    }

    // ---- BASELINE config A: Fibonacci 2^12 under app parameters (tests/test_gpu_parity_at_scale.py) ----
    dump(&out, "config_a_fib_2pow12_app", app_params_with_100_bits_security(12), FibFixture::new(0, 1, 1 << 12));
}
