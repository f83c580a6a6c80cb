//! `MultiStarkProvingKey` -> the flat `u32` "proving-key blob" of `ovm_hip_pk_load`
//! (include/openvm_hip.h has the format). Everything the device needs of a key is its parameters,
//! the symbolic DAG of every AIR (air_builders/symbolic/dag.rs:20-57), its interactions, and the
//! preprocessed traces; field constants go in canonical form.
use openvm_stark_backend::{
    air_builders::symbolic::{
        dag::SymbolicExpressionNode,
        symbolic_variable::Entry,
    },
    keygen::types::MultiStarkProvingKey,
    prover::MatrixDimensions,
};
use p3_field::PrimeField32;

use crate::prelude::{F, SC};

const MAGIC: u32 = 0x4F56_4D31;
const PREP_MAGIC: u32 = 0x5052_4550;

pub fn pk_blob(mpk: &MultiStarkProvingKey<SC>) -> Vec<u32> {
    let p = &mpk.params;
    let mut w: Vec<u32> = vec![
        MAGIC,
        p.l_skip as u32,
        p.n_stack as u32,
        p.w_stack as u32,
        p.log_blowup as u32,
        p.k_whir() as u32,
        p.whir.mu_pow_bits as u32,
        p.whir.query_phase_pow_bits as u32,
        p.whir.folding_pow_bits as u32,
        p.logup_pow_bits() as u32,
        mpk.max_constraint_degree as u32,
        p.whir.rounds.len() as u32,
    ];
    w.extend(p.whir.rounds.iter().map(|r| r.num_queries as u32));
    w.extend(mpk.vk_pre_hash.iter().map(|x: &F| x.as_canonical_u32()));
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
        let dag = &vk.symbolic_constraints;
        w.push(dag.constraints.nodes.len() as u32);
        for node in &dag.constraints.nodes {
            // (kind, entry, part_index, offset, x, y)
            let rec: [u32; 6] = match node {
                SymbolicExpressionNode::Variable(v) => {
                    let (entry, part, off) = match v.entry {
                        Entry::Preprocessed { offset } => (0, 0, offset),
                        Entry::Main { part_index, offset } => (1, part_index, offset),
                        Entry::Public => (2, 0, 0),
                        Entry::Challenge => unreachable!("challenge variables do not occur in AIR constraints"),
                    };
                    [0, entry, part as u32, off as u32, v.index as u32, 0]
                }
                SymbolicExpressionNode::IsFirstRow => [1, 0, 0, 0, 0, 0],
                SymbolicExpressionNode::IsLastRow => [2, 0, 0, 0, 0, 0],
                SymbolicExpressionNode::IsTransition => [3, 0, 0, 0, 0, 0],
                SymbolicExpressionNode::Constant(c) => [4, 0, 0, 0, c.as_canonical_u32(), 0],
                SymbolicExpressionNode::Add { left_idx, right_idx, .. } => [5, 0, 0, 0, *left_idx as u32, *right_idx as u32],
                SymbolicExpressionNode::Sub { left_idx, right_idx, .. } => [6, 0, 0, 0, *left_idx as u32, *right_idx as u32],
                SymbolicExpressionNode::Neg { idx, .. } => [7, 0, 0, 0, *idx as u32, 0],
                SymbolicExpressionNode::Mul { left_idx, right_idx, .. } => [8, 0, 0, 0, *left_idx as u32, *right_idx as u32],
            };
            w.extend(rec);
        }
        w.push(dag.constraints.constraint_idx.len() as u32);
        w.extend(dag.constraints.constraint_idx.iter().map(|&i| i as u32));
        w.push(dag.interactions.len() as u32);
        for it in &dag.interactions {
            w.extend([it.bus_index as u32, it.count_weight, it.count as u32, it.message.len() as u32]);
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
                w.extend(m.values.iter().map(|x: &F| x.as_canonical_u32()));
            }
        }
    }
    w
}
