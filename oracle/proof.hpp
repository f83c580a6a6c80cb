// ORACLE — test infrastructure only (see oracle/README.md).
//
// Proof structures and their canonical byte encoding (the bit-identity comparator), restated from
//   crates/stark-backend/src/proof.rs:20-213    structs
//   crates/stark-backend/src/proof.rs:226-421   Encode impls (CODEC_VERSION = 3)
//   crates/stark-backend/src/codec.rs:197-215   u32-LE lengths, canonical u32-LE field elements
#pragma once
#include "poseidon2.hpp"

namespace oracle {

struct TraceVData {
    size_t log_height = 0;
    std::vector<Digest> cached_commitments;
};

struct GkrLayerClaims {
    EF p_xi_0, p_xi_1, q_xi_0, q_xi_1;
};

struct GkrProof {
    Fp logup_pow_witness;
    EF q0_claim;
    std::vector<GkrLayerClaims> claims_per_layer;
    std::vector<std::vector<std::array<EF, 3>>> sumcheck_polys;
};

struct BatchConstraintProof {
    std::vector<EF> numerator_term_per_air, denominator_term_per_air;
    std::vector<EF> univariate_round_coeffs;
    std::vector<std::vector<EF>> sumcheck_round_polys;
    std::vector<std::vector<std::vector<EF>>> column_openings;
};

struct StackingProof {
    std::vector<EF> univariate_round_coeffs;
    std::vector<std::array<EF, 2>> sumcheck_round_polys;
    std::vector<std::vector<EF>> stacking_openings;
};

struct WhirProof {
    Fp mu_pow_witness;
    std::vector<std::array<EF, 2>> whir_sumcheck_polys;
    std::vector<Digest> codeword_commits;
    std::vector<EF> ood_values;
    std::vector<Fp> folding_pow_witnesses;
    std::vector<Fp> query_phase_pow_witnesses;
    // num_commits x num_queries x 2^k x stacking_width
    std::vector<std::vector<std::vector<std::vector<Fp>>>> initial_round_opened_rows;
    std::vector<std::vector<std::vector<Digest>>> initial_round_merkle_proofs;
    std::vector<std::vector<std::vector<EF>>> codeword_opened_values;
    std::vector<std::vector<std::vector<Digest>>> codeword_merkle_proofs;
    std::vector<EF> final_poly;
};

struct Proof {
    Digest common_main_commit;
    std::vector<bool> trace_present;
    std::vector<TraceVData> trace_vdata;  // indexed by air id; valid iff trace_present
    std::vector<std::vector<Fp>> public_values;
    GkrProof gkr_proof;
    BatchConstraintProof batch_constraint_proof;
    StackingProof stacking_proof;
    WhirProof whir_proof;
};

struct Writer {
    std::vector<uint8_t> buf;
    void u8(uint8_t x) { buf.push_back(x); }
    void u32(uint32_t x) {
        for (int i = 0; i < 4; i++) buf.push_back((uint8_t)(x >> (8 * i)));
    }
    void usize(size_t x) { u32((uint32_t)x); }
    void f(Fp x) { u32(x.v); }
    void ef(const EF &x) {
        for (int i = 0; i < 4; i++) f(x.c[i]);
    }
    void digest(const Digest &d) {
        for (int i = 0; i < 8; i++) f(d[i]);
    }
    void ef_slice(const std::vector<EF> &v) {
        usize(v.size());
        for (auto &x : v) ef(x);
    }
    void f_slice(const std::vector<Fp> &v) {
        usize(v.size());
        for (auto &x : v) f(x);
    }
    void digest_slice(const std::vector<Digest> &v) {
        usize(v.size());
        for (auto &x : v) digest(x);
    }
};

inline void encode(Writer &w, const GkrProof &p) {
    w.f(p.logup_pow_witness);
    w.ef(p.q0_claim);
    w.usize(p.claims_per_layer.size());
    for (auto &c : p.claims_per_layer) {
        w.ef(c.p_xi_0);
        w.ef(c.p_xi_1);
        w.ef(c.q_xi_0);
        w.ef(c.q_xi_1);
    }
    for (auto &round : p.sumcheck_polys)
        for (auto &arr : round)
            for (auto &x : arr) w.ef(x);
}

inline void encode(Writer &w, const BatchConstraintProof &p) {
    w.ef_slice(p.numerator_term_per_air);
    for (auto &x : p.denominator_term_per_air) w.ef(x);
    w.ef_slice(p.univariate_round_coeffs);
    size_t n_max = p.sumcheck_round_polys.size();
    w.usize(n_max);
    if (n_max > 0) {
        w.usize(p.sumcheck_round_polys[0].size());
        for (auto &r : p.sumcheck_round_polys)
            for (auto &x : r) w.ef(x);
    }
    for (auto &parts : p.column_openings) {
        w.usize(parts.size());
        for (auto &col : parts) w.ef_slice(col);
    }
}

inline void encode(Writer &w, const StackingProof &p) {
    w.ef_slice(p.univariate_round_coeffs);
    w.usize(p.sumcheck_round_polys.size());
    for (auto &a : p.sumcheck_round_polys)
        for (auto &x : a) w.ef(x);
    w.usize(p.stacking_openings.size());
    for (auto &o : p.stacking_openings) w.ef_slice(o);
}

inline void encode(Writer &w, const WhirProof &p) {
    w.f(p.mu_pow_witness);
    w.usize(p.whir_sumcheck_polys.size());
    for (auto &a : p.whir_sumcheck_polys)
        for (auto &x : a) w.ef(x);
    w.digest_slice(p.codeword_commits);
    for (auto &x : p.ood_values) w.ef(x);
    for (auto &x : p.folding_pow_witnesses) w.f(x);
    for (auto &x : p.query_phase_pow_witnesses) w.f(x);
    size_t num_commits = p.initial_round_opened_rows.size();
    w.usize(num_commits);
    size_t nq = p.initial_round_opened_rows[0].size();
    w.usize(nq);
    size_t num_whir_rounds = p.codeword_commits.size() + 1;
    if (nq > 0) {
        w.usize(p.initial_round_merkle_proofs[0][0].size());
        std::vector<size_t> widths;
        for (auto &commit_rows : p.initial_round_opened_rows) {
            size_t wd = 0;
            if (!commit_rows.empty() && !commit_rows[0].empty()) wd = commit_rows[0][0].size();
            widths.push_back(wd);
        }
        for (size_t wd : widths) w.usize(wd);
        for (auto &commit_rows : p.initial_round_opened_rows)
            for (auto &query_rows : commit_rows)
                for (auto &row : query_rows)
                    for (auto &x : row) w.f(x);
        for (auto &mps : p.initial_round_merkle_proofs)
            for (auto &proof : mps)
                for (auto &d : proof) w.digest(d);
    }
    for (auto &round : p.codeword_opened_values) {
        w.usize(round.size());
        for (auto &q : round)
            for (auto &x : q) w.ef(x);
    }
    size_t first_depth = 0;
    if (num_whir_rounds > 1 && nq > 0) first_depth = p.codeword_merkle_proofs[0][0].size();
    w.usize(first_depth);
    for (auto &round : p.codeword_merkle_proofs)
        for (auto &proof : round)
            for (auto &d : proof) w.digest(d);
    w.ef_slice(p.final_poly);
}

static constexpr uint32_t CODEC_VERSION = 3;

inline std::vector<uint8_t> encode_proof(const Proof &p) {
    Writer w;
    w.u32(CODEC_VERSION);
    w.digest(p.common_main_commit);
    size_t num_airs = p.trace_present.size();
    w.usize(num_airs);
    for (size_t i = 0; i < num_airs; i += 8) {
        uint8_t b = 0;
        for (size_t j = 0; j < 8 && i + j < num_airs; j++)
            if (p.trace_present[i + j]) b |= (uint8_t)(1u << j);
        w.u8(b);
    }
    for (size_t i = 0; i < num_airs; i++)
        if (p.trace_present[i]) {
            w.usize(p.trace_vdata[i].log_height);
            w.digest_slice(p.trace_vdata[i].cached_commitments);
        }
    w.usize(p.public_values.size());
    for (auto &pv : p.public_values) w.f_slice(pv);
    encode(w, p.gkr_proof);
    encode(w, p.batch_constraint_proof);
    encode(w, p.stacking_proof);
    encode(w, p.whir_proof);
    return w.buf;
}

}  // namespace oracle
