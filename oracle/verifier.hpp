// ORACLE — test infrastructure only (see oracle/README.md).
//
// CPU restatement of the reference VERIFIER, written from the verifier sources (not from the
// prover): an independent check that proofs produced by the oracle prover and by the HIP product
// satisfy every verifier equation, and that the byte encoding decodes.
//   crates/stark-backend/src/verifier/mod.rs:57-236                  verify
//   crates/stark-backend/src/verifier/batch_constraints.rs:52-387    verify_zerocheck_and_logup
//   crates/stark-backend/src/verifier/evaluator.rs:14-108            selectors at a point
//   crates/stark-backend/src/verifier/fractional_sumcheck_gkr.rs:40-234
//   crates/stark-backend/src/verifier/stacked_reduction.rs:28-239
//   crates/stark-backend/src/verifier/whir.rs:28-330
//   crates/stark-backend/src/proof.rs:423-706                        Decode impls
// Not restated: verify_proof_shape (verifier/proof_shape.rs) beyond what decoding needs, and
// trace-height linear constraints (keygen data outside the hot path).
#pragma once
#include "air.hpp"
#include "proof.hpp"

namespace oracle {

struct Reader {
    const uint8_t *p;
    size_t n, i = 0;
    uint8_t u8() {
        if (i >= n) throw std::runtime_error("proof truncated");
        return p[i++];
    }
    uint32_t u32() {
        uint32_t x = 0;
        for (int k = 0; k < 4; k++) x |= (uint32_t)u8() << (8 * k);
        return x;
    }
    size_t usize() { return u32(); }
    Fp f() {
        uint32_t x = u32();
        if (x >= P) throw std::runtime_error("non-canonical field element");
        return Fp((uint64_t)x);
    }
    EF ef() {
        EF r;
        for (int k = 0; k < 4; k++) r.c[k] = f();
        return r;
    }
    Digest digest() {
        Digest d;
        for (int k = 0; k < 8; k++) d[k] = f();
        return d;
    }
    std::vector<EF> ef_n(size_t k) {
        std::vector<EF> v(k);
        for (auto &x : v) x = ef();
        return v;
    }
    std::vector<EF> ef_vec() { return ef_n(usize()); }
    std::vector<Fp> f_n(size_t k) {
        std::vector<Fp> v(k);
        for (auto &x : v) x = f();
        return v;
    }
    std::vector<Digest> digest_n(size_t k) {
        std::vector<Digest> v(k);
        for (auto &x : v) x = digest();
        return v;
    }
};

inline Proof decode_proof(const uint8_t *bytes, size_t len) {
    Reader r{bytes, len};
    Proof p;
    if (r.u32() != CODEC_VERSION) throw std::runtime_error("CODEC_VERSION mismatch");
    p.common_main_commit = r.digest();
    size_t num_airs = r.usize();
    std::vector<uint8_t> bitmap((num_airs + 7) / 8);
    for (auto &b : bitmap) b = r.u8();
    p.trace_present.assign(num_airs, false);
    p.trace_vdata.resize(num_airs);
    for (size_t a = 0; a < num_airs; a++)
        if (bitmap[a / 8] >> (a % 8) & 1) {
            p.trace_present[a] = true;
            p.trace_vdata[a].log_height = r.usize();
            p.trace_vdata[a].cached_commitments = r.digest_n(r.usize());
        }
    size_t num_pvs = r.usize();
    for (size_t i = 0; i < num_pvs; i++) p.public_values.push_back(r.f_n(r.usize()));
    // gkr
    p.gkr_proof.logup_pow_witness = r.f();
    p.gkr_proof.q0_claim = r.ef();
    size_t layers = r.usize();
    for (size_t i = 0; i < layers; i++) {
        GkrLayerClaims c;
        c.p_xi_0 = r.ef();
        c.p_xi_1 = r.ef();
        c.q_xi_0 = r.ef();
        c.q_xi_1 = r.ef();
        p.gkr_proof.claims_per_layer.push_back(c);
    }
    for (size_t j = 1; j < layers; j++) {
        std::vector<std::array<EF, 3>> round(j);
        for (auto &a : round)
            for (auto &x : a) x = r.ef();
        p.gkr_proof.sumcheck_polys.push_back(round);
    }
    // batch constraints
    BatchConstraintProof &b = p.batch_constraint_proof;
    b.numerator_term_per_air = r.ef_vec();
    size_t present = b.numerator_term_per_air.size();
    b.denominator_term_per_air = r.ef_n(present);
    b.univariate_round_coeffs = r.ef_vec();
    size_t n_max = r.usize();
    if (n_max > 0) {
        size_t d1 = r.usize();
        for (size_t i = 0; i < n_max; i++) b.sumcheck_round_polys.push_back(r.ef_n(d1));
    }
    for (size_t t = 0; t < present; t++) {
        size_t parts = r.usize();
        std::vector<std::vector<EF>> ps;
        for (size_t k = 0; k < parts; k++) ps.push_back(r.ef_vec());
        b.column_openings.push_back(ps);
    }
    // stacking
    StackingProof &s = p.stacking_proof;
    s.univariate_round_coeffs = r.ef_vec();
    size_t rounds = r.usize();
    for (size_t i = 0; i < rounds; i++) {
        std::array<EF, 2> a{r.ef(), EF()};
        a[1] = r.ef();
        s.sumcheck_round_polys.push_back(a);
    }
    size_t nopen = r.usize();
    for (size_t i = 0; i < nopen; i++) s.stacking_openings.push_back(r.ef_vec());
    // whir
    WhirProof &w = p.whir_proof;
    w.mu_pow_witness = r.f();
    size_t nsc = r.usize();
    for (size_t i = 0; i < nsc; i++) {
        std::array<EF, 2> a;
        a[0] = r.ef();
        a[1] = r.ef();
        w.whir_sumcheck_polys.push_back(a);
    }
    w.codeword_commits = r.digest_n(r.usize());
    size_t nwr = w.codeword_commits.size() + 1;
    if (nsc % nwr) throw std::runtime_error("num_whir_sumcheck_rounds not a multiple of rounds");
    size_t k_whir = nsc / nwr, kexp = size_t(1) << k_whir;
    w.ood_values = r.ef_n(nwr - 1);
    w.folding_pow_witnesses = r.f_n(nsc);
    w.query_phase_pow_witnesses = r.f_n(nwr);
    size_t num_commits = r.usize();
    if (!num_commits) throw std::runtime_error("num_commits must be nonzero");
    size_t nq = r.usize();
    size_t depth = 0;
    std::vector<size_t> widths(num_commits, 0);
    if (nq > 0) {
        depth = r.usize();
        for (auto &x : widths) x = r.usize();
    }
    for (size_t c = 0; c < num_commits; c++) {
        std::vector<std::vector<std::vector<Fp>>> rows(nq);
        for (auto &q : rows)
            for (size_t j = 0; j < kexp; j++) q.push_back(r.f_n(widths[c]));
        w.initial_round_opened_rows.push_back(rows);
    }
    for (size_t c = 0; c < num_commits; c++) {
        std::vector<std::vector<Digest>> mp(nq);
        for (auto &q : mp) q = r.digest_n(depth);
        w.initial_round_merkle_proofs.push_back(mp);
    }
    for (size_t i = 0; i + 1 < nwr; i++) {
        size_t q = r.usize();
        std::vector<std::vector<EF>> ov(q);
        for (auto &x : ov) x = r.ef_n(kexp);
        w.codeword_opened_values.push_back(ov);
    }
    depth = r.usize();
    for (auto &ov : w.codeword_opened_values) {
        std::vector<std::vector<Digest>> mp(ov.size());
        for (auto &x : mp) x = r.digest_n(depth);
        w.codeword_merkle_proofs.push_back(mp);
        depth--;
    }
    w.final_poly = r.ef_vec();
    if (r.i != len) throw std::runtime_error("trailing bytes after decoded proof");
    return p;
}

// --- helpers ---------------------------------------------------------------------------------
inline EF interpolate_linear_at_01(EF e0, EF e1, EF x) { return (e1 - e0) * x + e0; }
inline EF interpolate_quadratic_at_012(EF e0, EF e1, EF e2, EF x) {
    EF s1 = e1 - e0, s2 = e2 - e1;
    EF p = (s2 - s1).halve();
    EF q = s1 - p;
    return (p * x + q) * x + e0;
}
inline EF interpolate_cubic_at_0123(EF e0, EF e1, EF e2, EF e3, EF x) {
    Fp inv6 = Fp(6).inv();
    EF s1 = e1 - e0, s2 = e2 - e0, s3 = e3 - e0;
    EF d3 = s3 - (s2 - s1) * Fp(3);
    EF p = d3 * inv6;
    EF q = (s2 - d3).halve() - s1;
    EF rr = s1 - p - q;
    return ((p * x + q) * x + rr) * x + e0;
}
inline EF eval_mobius_eq_mle(const std::vector<EF> &u, const std::vector<EF> &x) {
    EF acc = EF::one();
    for (size_t i = 0; i < u.size(); i++) {
        EF w0 = EF::one() - u[i].dbl();
        acc = acc * (w0 * (EF::one() - x[i]) + u[i] * x[i]);
    }
    return acc;
}
inline EF eval_mle_evals_at_point(std::vector<EF> evals, const std::vector<EF> &x) {
    size_t len = evals.size();
    for (size_t k = x.size(); k-- > 0;) {
        len >>= 1;
        for (size_t i = 0; i < len; i++) evals[i] = evals[i] * (EF::one() - x[k]) + evals[len + i] * x[k];
    }
    return evals[0];
}
inline EF progression_exp_2(EF m, unsigned l) {
    EF pow = m, sum = EF::one();
    for (unsigned i = 0; i < l; i++) {
        sum = sum * (EF::one() + pow);
        pow = pow * pow;
    }
    return sum;
}

struct VerifyError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// fractional_sumcheck_gkr.rs:40-140
inline void verify_gkr(const GkrProof &proof, Transcript &ts, size_t total_rounds, EF &numer, EF &denom,
                       std::vector<EF> &gkr_r) {
    if (proof.claims_per_layer.size() != total_rounds) throw VerifyError("IncorrectLayerCount");
    if (proof.sumcheck_polys.size() != total_rounds - 1) throw VerifyError("IncorrectSumcheckPolyCount");
    ts.observe_ext(proof.q0_claim);
    auto observe = [&](const GkrLayerClaims &c) {
        ts.observe_ext(c.p_xi_0);
        ts.observe_ext(c.q_xi_0);
        ts.observe_ext(c.p_xi_1);
        ts.observe_ext(c.q_xi_1);
    };
    const GkrLayerClaims *lc = &proof.claims_per_layer[0];
    observe(*lc);
    EF p_cross = lc->p_xi_0 * lc->q_xi_1 + lc->p_xi_1 * lc->q_xi_0, q_cross = lc->q_xi_0 * lc->q_xi_1;
    if (!p_cross.is_zero()) throw VerifyError("GKR ZeroCheckFailed");
    if (q_cross != proof.q0_claim) throw VerifyError("GKR RootConsistencyCheckFailed");
    EF mu = ts.sample_ext();
    numer = interpolate_linear_at_01(lc->p_xi_0, lc->p_xi_1, mu);
    denom = interpolate_linear_at_01(lc->q_xi_0, lc->q_xi_1, mu);
    gkr_r = {mu};
    for (size_t round = 1; round < total_rounds; round++) {
        EF lambda = ts.sample_ext();
        EF claim = numer + lambda * denom;
        const auto &polys = proof.sumcheck_polys[round - 1];
        if (polys.size() != round) throw VerifyError("IncorrectSubroundCount");
        std::vector<EF> r_prime;
        EF eq = EF::one();
        for (size_t sr = 0; sr < round; sr++) {
            for (auto &e : polys[sr]) ts.observe_ext(e);
            EF ri = ts.sample_ext();
            r_prime.push_back(ri);
            EF ev0 = claim - polys[sr][0];
            claim = interpolate_cubic_at_0123(ev0, polys[sr][0], polys[sr][1], polys[sr][2], ri);
            EF xi = gkr_r[sr];
            eq = eq * (xi * ri + (EF::one() - xi) * (EF::one() - ri));
        }
        lc = &proof.claims_per_layer[round];
        observe(*lc);
        p_cross = lc->p_xi_0 * lc->q_xi_1 + lc->p_xi_1 * lc->q_xi_0;
        q_cross = lc->q_xi_0 * lc->q_xi_1;
        if ((p_cross + lambda * q_cross) * eq != claim) throw VerifyError("GKR LayerConsistencyCheckFailed");
        mu = ts.sample_ext();
        numer = interpolate_linear_at_01(lc->p_xi_0, lc->p_xi_1, mu);
        denom = interpolate_linear_at_01(lc->q_xi_0, lc->q_xi_1, mu);
        gkr_r.clear();
        gkr_r.push_back(mu);
        gkr_r.insert(gkr_r.end(), r_prime.begin(), r_prime.end());
    }
}

inline std::vector<std::pair<EF, EF>> openings_by_rot(const std::vector<EF> &o, bool need_rot) {
    std::vector<std::pair<EF, EF>> v;
    if (need_rot)
        for (size_t i = 0; i + 1 < o.size(); i += 2) v.push_back({o[i], o[i + 1]});
    else
        for (auto &x : o) v.push_back({x, EF::zero()});
    return v;
}

// batch_constraints.rs:52-387
inline std::vector<EF> verify_zerocheck_and_logup(Transcript &ts, const MultiPk &mvk, const Proof &proof,
                                                  const std::vector<size_t> &trace_id_to_air_id,
                                                  const std::vector<int> &n_per_trace,
                                                  const std::vector<Fp> &omega_skip_pows) {
    const unsigned l_skip = mvk.params.l_skip;
    const BatchConstraintProof &bp = proof.batch_constraint_proof;
    const GkrProof &gkr = proof.gkr_proof;
    if (!ts.check_witness(mvk.params.logup_pow_bits, gkr.logup_pow_witness)) throw VerifyError("InvalidLogupPowWitness");
    EF alpha = ts.sample_ext(), beta = ts.sample_ext();
    uint64_t total_interactions = 0;
    for (size_t t = 0; t < trace_id_to_air_id.size(); t++) {
        unsigned n_lift = (unsigned)std::max(n_per_trace[t], 0);
        total_interactions += (uint64_t)mvk.per_air[trace_id_to_air_id[t]].vk.interactions.size() << (l_skip + n_lift);
    }
    unsigned n_logup = calculate_n_logup(l_skip, total_interactions);
    std::vector<EF> xi;
    EF p_xi = EF::zero(), q_xi = alpha;
    if (total_interactions > 0)
        verify_gkr(gkr, ts, l_skip + n_logup, p_xi, q_xi, xi);
    else if (gkr.q0_claim != EF::one())
        throw VerifyError("InvalidZeroRoundValue");
    int nm = 0;
    for (int n : n_per_trace) nm = std::max(nm, n);
    unsigned n_max = (unsigned)nm, n_global = std::max(n_max, n_logup);
    while (xi.size() != l_skip + n_global) xi.push_back(ts.sample_ext());
    EF lambda = ts.sample_ext();
    for (size_t i = 0; i < bp.numerator_term_per_air.size(); i++) {
        p_xi -= bp.numerator_term_per_air[i];
        q_xi -= bp.denominator_term_per_air[i];
        ts.observe_ext(bp.numerator_term_per_air[i]);
        ts.observe_ext(bp.denominator_term_per_air[i]);
    }
    if (!p_xi.is_zero()) throw VerifyError("GkrNumeratorMismatch");
    if (q_xi != alpha) throw VerifyError("GkrDenominatorMismatch");
    EF mu = ts.sample_ext();
    EF sum_claim = EF::zero(), cur_mu = EF::one();
    for (size_t i = 0; i < bp.numerator_term_per_air.size(); i++) {
        sum_claim += bp.numerator_term_per_air[i] * cur_mu;
        cur_mu *= mu;
        sum_claim += bp.denominator_term_per_air[i] * cur_mu;
        cur_mu *= mu;
    }
    for (auto &c : bp.univariate_round_coeffs) ts.observe_ext(c);
    const unsigned s_deg = mvk.params.max_constraint_degree + 1;
    EF r_0 = ts.sample_ext();
    const size_t N = size_t(1) << l_skip;
    if (bp.univariate_round_coeffs.size() != (size_t)s_deg * (N - 1) + 1) throw VerifyError("s_0 length");
    EF sum_dom = EF::zero();
    for (size_t j = 0; j < bp.univariate_round_coeffs.size(); j += N) sum_dom += bp.univariate_round_coeffs[j];
    sum_dom = sum_dom * Fp((uint64_t)N);
    if (sum_claim != sum_dom) throw VerifyError("SumClaimMismatch");
    EF cur_sum = horner_eval(bp.univariate_round_coeffs, r_0);
    std::vector<EF> rs{r_0};
    if (bp.sumcheck_round_polys.size() != n_max) throw VerifyError("sumcheck_round_polys length");
    for (unsigned round = 0; round < n_max; round++) {
        const auto &ev = bp.sumcheck_round_polys[round];
        if (ev.size() != s_deg) throw VerifyError("round poly length");
        for (auto &e : ev) ts.observe_ext(e);
        std::vector<EF> all{cur_sum - ev[0]};
        all.insert(all.end(), ev.begin(), ev.end());
        std::vector<Fp> fact(s_deg + 1, Fp::one());
        for (unsigned i = 1; i <= s_deg; i++) fact[i] = fact[i - 1] * Fp((uint64_t)i);
        EF r = ts.sample_ext();
        std::vector<EF> pref(s_deg + 1, EF::one()), suf(s_deg + 1, EF::one());
        for (unsigned i = 0; i < s_deg; i++) {
            pref[i + 1] = pref[i] * (r - EF::from_u64(i));
            suf[i + 1] = suf[i] * (EF::from_u64(s_deg - i) - r);
        }
        cur_sum = EF::zero();
        for (unsigned i = 0; i <= s_deg; i++)
            cur_sum += all[i] * pref[i] * suf[s_deg - i] * fact[i].inv() * fact[s_deg - i].inv();
        rs.push_back(r);
    }
    size_t stacked_idx = 0;
    std::vector<std::vector<EF>> eq_3b(n_per_trace.size());
    for (size_t t = 0; t < n_per_trace.size(); t++) {
        const AirVk &vk = mvk.per_air[trace_id_to_air_id[t]].vk;
        unsigned n_lift = (unsigned)std::max(n_per_trace[t], 0);
        for (size_t i = 0; i < vk.interactions.size(); i++) {
            size_t b_int = stacked_idx >> (l_skip + n_lift);
            std::vector<Fp> b(n_logup - n_lift);
            for (auto &x : b) {
                x = Fp(b_int & 1);
                b_int >>= 1;
            }
            stacked_idx += size_t(1) << (l_skip + n_lift);
            eq_3b[t].push_back(eval_eq_mle(std::vector<EF>(xi.begin() + l_skip + n_lift, xi.begin() + l_skip + n_logup), b));
        }
    }
    std::vector<EF> eq_ns(n_max + 1, EF::one()), eq_sharp_ns(n_max + 1, EF::one());
    eq_ns[0] = eval_eq_uni(l_skip, xi[0], r_0);
    eq_sharp_ns[0] = eval_eq_sharp_uni(omega_skip_pows, std::vector<EF>(xi.begin(), xi.begin() + l_skip), r_0);
    for (size_t i = 1; i < rs.size(); i++) {
        EF e = eval_eq_mle(std::vector<EF>{xi[l_skip + i - 1]}, std::vector<EF>{rs[i]});
        eq_ns[i] = eq_ns[i - 1] * e;
        eq_sharp_ns[i] = eq_sharp_ns[i - 1] * e;
    }
    EF r_rev = rs[n_max];
    for (size_t i = n_max; i-- > 0;) {
        eq_ns[i] *= r_rev;
        eq_sharp_ns[i] *= r_rev;
        r_rev *= rs[i];
    }
    if (bp.column_openings.size() != n_per_trace.size()) throw VerifyError("column_openings length");
    for (size_t t = 0; t < bp.column_openings.size(); t++) {
        bool rot = mvk.per_air[trace_id_to_air_id[t]].vk.need_rot;
        for (auto &[c, cr] : openings_by_rot(bp.column_openings[t][0], rot)) {
            ts.observe_ext(c);
            ts.observe_ext(cr);
        }
    }
    std::vector<EF> interactions_evals, constraints_evals;
    for (size_t t = 0; t < bp.column_openings.size(); t++) {
        size_t air_idx = trace_id_to_air_id[t];
        const AirPk &pk = mvk.per_air[air_idx];
        const AirVk &vk = pk.vk;
        int n = n_per_trace[t];
        unsigned n_lift = (unsigned)std::max(n, 0);
        bool rot = vk.need_rot;
        const auto &ao = bp.column_openings[t];
        for (size_t p = 1; p < ao.size(); p++)
            for (auto &[c, cr] : openings_by_rot(ao[p], rot)) {
                ts.observe_ext(c);
                ts.observe_ext(cr);
            }
        bool has_pre = vk.has_preprocessed;
        // row parts in RowView order: preprocessed, cached..., common; (local, next) each
        std::vector<std::vector<EF>> parts;
        auto push_part = [&](const std::vector<EF> &o) {
            auto pairs = openings_by_rot(o, rot);
            std::vector<EF> loc, nxt;
            for (auto &[c, cr] : pairs) {
                loc.push_back(c);
                nxt.push_back(cr);
            }
            parts.push_back(loc);
            if (rot) parts.push_back(nxt);
        };
        if (has_pre) push_part(ao[1]);
        for (size_t p = 1 + (has_pre ? 1 : 0); p < ao.size(); p++) push_part(ao[p]);
        push_part(ao[0]);
        unsigned l = l_skip;
        std::vector<EF> rs_n;
        Fp norm = Fp::one();
        if (n < 0) {
            l = l_skip + n;
            rs_n = {rs[0].exp_power_of_2((unsigned)(-n))};
            norm = Fp((uint64_t)1 << (unsigned)(-n)).inv();
        } else {
            rs_n.assign(rs.begin(), rs.begin() + n + 1);
        }
        Fp omega = two_adic_generator(l);
        EF inv = EF(Fp((uint64_t)1 << l).inv());
        EF is_first = inv * progression_exp_2(rs_n[0], l), is_last = inv * progression_exp_2(rs_n[0] * omega, l);
        for (size_t i = 1; i < rs_n.size(); i++) {
            is_first = is_first * (EF::one() - rs_n[i]);
            is_last = is_last * rs_n[i];
        }
        RowView<EF> row;
        row.is_first = is_first;
        row.is_last = is_last;
        row.is_transition = EF::one() - is_last;
        row.parts = &parts;
        row.needs_next = rot;
        row.has_preprocessed = has_pre;
        if (proof.public_values[air_idx].size() != vk.num_public_values) throw VerifyError("public values length");
        row.public_values = &proof.public_values[air_idx];
        std::vector<EF> nodes = eval_nodes(vk, row);
        EF expr = EF::zero(), lp = EF::one();
        for (size_t i = 0; i < vk.constraint_idx.size(); i++) {
            expr += nodes[vk.constraint_idx[i]] * lp;
            lp *= lambda;
        }
        constraints_evals.push_back(eq_ns[n_lift] * expr);
        EF num = EF::zero(), den = EF::zero();
        for (size_t i = 0; i < vk.interactions.size(); i++) {
            const Interaction &it = vk.interactions[i];
            EF d = EF::zero(), bp_ = EF::one();
            for (uint32_t m : it.message) {
                d += nodes[m] * bp_;
                bp_ *= beta;
            }
            d += EF::from_u64((uint64_t)it.bus_index + 1) * bp_;
            num += eq_3b[t][i] * nodes[it.count];
            den += eq_3b[t][i] * d;
        }
        interactions_evals.push_back(num * norm * eq_sharp_ns[n_lift]);
        interactions_evals.push_back(den * eq_sharp_ns[n_lift]);
    }
    EF evaluated = EF::zero(), mp = EF::one();
    for (auto &x : interactions_evals) {
        evaluated += x * mp;
        mp *= mu;
    }
    for (auto &x : constraints_evals) {
        evaluated += x * mp;
        mp *= mu;
    }
    if (cur_sum != evaluated) throw VerifyError("BatchConstraint InconsistentClaims");
    return rs;
}

// stacked_reduction.rs:28-239
inline std::vector<EF> verify_stacked_reduction(Transcript &ts, const StackingProof &proof,
                                                const std::vector<StackedLayout> &layouts,
                                                const std::vector<std::vector<bool>> &need_rot_per_commit,
                                                unsigned l_skip, unsigned n_stack,
                                                const std::vector<std::vector<std::vector<EF>>> &column_openings,
                                                const std::vector<EF> &r) {
    const size_t N = size_t(1) << l_skip;
    size_t lambda_idx = 0;
    std::vector<std::vector<std::pair<size_t, bool>>> lam(layouts.size());
    for (size_t c = 0; c < layouts.size(); c++)
        for (auto &sc : layouts[c].sorted_cols) lam[c].push_back({lambda_idx++, need_rot_per_commit[c][sc.mat_idx]});
    std::vector<std::pair<EF, EF>> t_claims;
    for (size_t t = 0; t < column_openings.size(); t++)
        for (auto &pr : openings_by_rot(column_openings[t][0], need_rot_per_commit[0][t])) t_claims.push_back(pr);
    size_t commit_idx = 1;
    for (auto &parts : column_openings)
        for (size_t p = 1; p < parts.size(); p++) {
            for (auto &pr : openings_by_rot(parts[p], need_rot_per_commit[commit_idx][0])) t_claims.push_back(pr);
            commit_idx++;
        }
    if (t_claims.size() != lambda_idx) throw VerifyError("t_claims length");
    EF lambda = ts.sample_ext();
    std::vector<EF> l2 = powers(lambda * lambda, t_claims.size());
    EF s_0 = EF::zero();
    for (size_t i = 0; i < t_claims.size(); i++) s_0 += (t_claims[i].first + t_claims[i].second * lambda) * l2[i];
    EF s_0_sum = EF::zero();
    for (size_t j = 0; j < proof.univariate_round_coeffs.size(); j += N) s_0_sum += proof.univariate_round_coeffs[j];
    s_0_sum = s_0_sum * Fp((uint64_t)N);
    if (s_0 != s_0_sum) throw VerifyError("Stacking S0Mismatch");
    for (auto &c : proof.univariate_round_coeffs) ts.observe_ext(c);
    std::vector<EF> u(n_stack + 1);
    u[0] = ts.sample_ext();
    EF claim = horner_eval(proof.univariate_round_coeffs, u[0]);
    if (proof.sumcheck_round_polys.size() != n_stack) throw VerifyError("stacking rounds");
    for (unsigned j = 1; j <= n_stack; j++) {
        EF s1 = proof.sumcheck_round_polys[j - 1][0], s2 = proof.sumcheck_round_polys[j - 1][1];
        ts.observe_ext(s1);
        ts.observe_ext(s2);
        u[j] = ts.sample_ext();
        claim = interpolate_quadratic_at_012(claim - s1, s1, s2, u[j]);
    }
    EF final_sum = EF::zero();
    if (proof.stacking_openings.size() != layouts.size()) throw VerifyError("stacking_openings length");
    std::vector<std::vector<EF>> q_coeffs;
    for (auto &v : proof.stacking_openings) q_coeffs.push_back(std::vector<EF>(v.size(), EF::zero()));
    for (size_t c = 0; c < layouts.size(); c++)
        for (size_t ci = 0; ci < layouts[c].sorted_cols.size(); ci++) {
            const StackedSlice &s = layouts[c].sorted_cols[ci].slice;
            auto [li, rot] = lam[c][ci];
            int n = (int)s.log_height - (int)l_skip;
            unsigned n_lift = (unsigned)std::max(n, 0);
            std::vector<Fp> b;
            for (unsigned j = l_skip + n_lift; j < l_skip + n_stack; j++) b.push_back(Fp((s.row_idx >> j) & 1));
            EF eq_mle = eval_eq_mle(std::vector<EF>(u.begin() + n_lift + 1, u.end()), b);
            EF ind = eval_in_uni(l_skip, n, u[0]);
            unsigned l = l_skip;
            std::vector<EF> rs_n;
            if (n < 0) {
                l = l_skip + n;
                rs_n = {r[0].exp_power_of_2((unsigned)(-n))};
            } else {
                rs_n.assign(r.begin(), r.begin() + n_lift + 1);
            }
            std::vector<EF> us(u.begin(), u.begin() + n_lift + 1);
            EF eq_prism = eval_eq_uni(l, us[0], rs_n[0]) *
                          eval_eq_mle(std::vector<EF>(us.begin() + 1, us.end()), std::vector<EF>(rs_n.begin() + 1, rs_n.end()));
            EF batched = l2[li] * eq_prism;
            if (rot) batched += l2[li] * lambda * eval_rot_kernel_prism(l, us, rs_n);
            if (s.col_idx >= q_coeffs[c].size()) throw VerifyError("stacking opening width");
            q_coeffs[c][s.col_idx] += eq_mle * batched * ind;
        }
    for (size_t c = 0; c < q_coeffs.size(); c++)
        for (size_t j = 0; j < q_coeffs[c].size(); j++) {
            ts.observe_ext(proof.stacking_openings[c][j]);
            final_sum += q_coeffs[c][j] * proof.stacking_openings[c][j];
        }
    if (claim != final_sum) throw VerifyError("Stacking FinalSumMismatch");
    return u;
}

// whir.rs:282-310
inline EF binary_k_fold(std::vector<EF> values, const std::vector<EF> &alphas, Fp x) {
    size_t n = values.size(), k = alphas.size();
    Fp omega_k = two_adic_generator((unsigned)k), omega_k_inv = omega_k.inv();
    std::vector<Fp> tw = powers(omega_k, size_t(1) << (k - 1)), inv_tw = powers(omega_k_inv, size_t(1) << (k - 1));
    Fp x_pow = x, x_inv_pow = x.inv();
    for (size_t j = 0; j < k; j++) {
        size_t m = n >> (j + 1);
        for (size_t i = 0; i < m; i++) {
            Fp t = tw[i << j] * x_pow, t_inv = inv_tw[i << j] * x_inv_pow;
            values[i] += (alphas[j] - t) * (values[i] - values[m + i]) * t_inv.halve();
        }
        x_pow = x_pow.square();
        x_inv_pow = x_inv_pow.square();
    }
    return values[0];
}

inline void merkle_verify(const Digest &root, uint32_t idx, Digest cur, const std::vector<Digest> &path) {
    for (auto &sib : path) {
        cur = (idx & 1) == 0 ? compress(cur, sib) : compress(sib, cur);
        idx >>= 1;
    }
    if (cur != root) throw VerifyError("MerkleVerify");
}
inline Digest tree_compress(std::vector<Digest> h) {
    while (h.size() > 1) {
        std::vector<Digest> nx;
        for (size_t i = 0; i + 1 < h.size(); i += 2) nx.push_back(compress(h[i], h[i + 1]));
        h = nx;
    }
    return h[0];
}

// whir.rs:28-262
inline void verify_whir(Transcript &ts, const SystemParams &params, const WhirProof &wp,
                        const std::vector<std::vector<EF>> &stacking_openings,
                        const std::vector<Digest> &commitments, const std::vector<EF> &u) {
    if (!ts.check_witness(params.mu_pow_bits, wp.mu_pow_witness)) throw VerifyError("MuPoWInvalid");
    EF mu = ts.sample_ext();
    const unsigned m = params.l_skip + params.n_stack, k_whir = params.k_whir;
    const size_t num_whir_rounds = params.num_whir_rounds();
    unsigned log_rs = m + params.log_blowup;
    size_t total_w = 0;
    for (auto &v : stacking_openings) total_w += v.size();
    std::vector<EF> mu_pows = powers(mu, total_w);
    EF claim = EF::zero();
    {
        size_t j = 0;
        for (auto &v : stacking_openings)
            for (auto &o : v) claim += mu_pows[j++] * o;
    }
    if (wp.query_phase_pow_witnesses.size() != num_whir_rounds) throw VerifyError("whir rounds");
    if (wp.whir_sumcheck_polys.size() != num_whir_rounds * k_whir) throw VerifyError("whir sumcheck rounds");
    std::vector<EF> gammas, z0s, alphas;
    std::vector<std::vector<Fp>> zs;
    size_t sc_i = 0;
    for (size_t wr = 0; wr < num_whir_rounds; wr++) {
        bool initial = wr == 0, final_round = wr + 1 == num_whir_rounds;
        std::vector<EF> alphas_round;
        for (unsigned k = 0; k < k_whir; k++, sc_i++) {
            EF ev1 = wp.whir_sumcheck_polys[sc_i][0], ev2 = wp.whir_sumcheck_polys[sc_i][1];
            ts.observe_ext(ev1);
            ts.observe_ext(ev2);
            if (!ts.check_witness(params.folding_pow_bits, wp.folding_pow_witnesses[sc_i])) throw VerifyError("FoldingPoWInvalid");
            EF alpha = ts.sample_ext();
            alphas_round.push_back(alpha);
            claim = interpolate_quadratic_at_012(claim - ev1, ev1, ev2, alpha);
        }
        bool have_y0 = false;
        EF y0;
        if (final_round) {
            for (auto &c : wp.final_poly) ts.observe_ext(c);
        } else {
            ts.observe_commit(wp.codeword_commits[wr]);
            z0s.push_back(ts.sample_ext());
            y0 = wp.ood_values[wr];
            ts.observe_ext(y0);
            have_y0 = true;
        }
        if (!ts.check_witness(params.query_phase_pow_bits, wp.query_phase_pow_witnesses[wr])) throw VerifyError("QueryPhasePoWInvalid");
        size_t nq = params.whir_num_queries[wr];
        std::vector<uint64_t> idxs;
        for (size_t q = 0; q < nq; q++) idxs.push_back(ts.sample_bits(log_rs - k_whir));
        std::vector<Fp> zs_round;
        std::vector<EF> ys_round;
        Fp omega = two_adic_generator(log_rs);
        for (size_t q = 0; q < nq; q++) {
            uint64_t index = idxs[q];
            Fp zi_root = omega.pow(index), zi = zi_root.exp_power_of_2(k_whir);
            EF yi;
            if (initial) {
                std::vector<EF> vals(size_t(1) << k_whir, EF::zero());
                size_t mj = 0;
                for (size_t c = 0; c < commitments.size(); c++) {
                    const auto &rows = wp.initial_round_opened_rows[c][q];
                    if (rows.size() != vals.size()) throw VerifyError("opened rows count");
                    std::vector<Digest> leaves;
                    for (auto &row : rows) leaves.push_back(hash_slice(row));
                    merkle_verify(commitments[c], (uint32_t)index, tree_compress(leaves), wp.initial_round_merkle_proofs[c][q]);
                    size_t width = stacking_openings[c].size();
                    for (size_t cc = 0; cc < width; cc++, mj++)
                        for (size_t j = 0; j < vals.size(); j++) vals[j] += mu_pows[mj] * rows[j][cc];
                }
                yi = binary_k_fold(vals, alphas_round, zi_root);
            } else {
                const auto &ov = wp.codeword_opened_values[wr - 1][q];
                std::vector<Digest> leaves;
                for (auto &x : ov) leaves.push_back(hash_slice(x.c, 4));
                merkle_verify(wp.codeword_commits[wr - 1], (uint32_t)index, tree_compress(leaves), wp.codeword_merkle_proofs[wr - 1][q]);
                yi = binary_k_fold(ov, alphas_round, zi_root);
            }
            zs_round.push_back(zi);
            ys_round.push_back(yi);
        }
        EF gamma = ts.sample_ext();
        if (have_y0) claim += y0 * gamma;
        EF gp = gamma * gamma;
        for (auto &yi : ys_round) {
            claim += yi * gp;
            gp *= gamma;
        }
        gammas.push_back(gamma);
        zs.push_back(zs_round);
        alphas.insert(alphas.end(), alphas_round.begin(), alphas_round.end());
        log_rs -= 1;
    }
    size_t t = (size_t)k_whir * num_whir_rounds;
    if (wp.final_poly.size() != (size_t(1) << (m - t))) throw VerifyError("FinalPolyDegree");
    EF prefix = eval_mobius_eq_mle(std::vector<EF>(u.begin(), u.begin() + t), alphas);
    EF suffix = eval_mle_evals_at_point(wp.final_poly, std::vector<EF>(u.begin() + t, u.end()));
    EF acc = prefix * suffix;
    size_t j = k_whir;
    for (size_t i = 0; i < num_whir_rounds; i++) {
        std::vector<EF> alpha_slc(alphas.begin() + j, alphas.begin() + t);
        size_t slc_len = (t - j) + 1;
        if (i + 1 != num_whir_rounds) {
            std::vector<EF> zp(slc_len);
            EF cur = z0s[i];
            for (auto &z : zp) {
                z = cur;
                cur = cur.square();
            }
            EF zmax = zp.back();
            zp.pop_back();
            acc += gammas[i] * eval_eq_mle(alpha_slc, zp) * horner_eval(wp.final_poly, zmax);
        }
        EF gp = gammas[i] * gammas[i];
        for (Fp zi : zs[i]) {
            std::vector<Fp> zp(slc_len);
            Fp cur = zi;
            for (auto &z : zp) {
                z = cur;
                cur = cur.square();
            }
            Fp zmax = zp.back();
            zp.pop_back();
            acc += gp * eval_eq_mle(alpha_slc, zp) * horner_eval(wp.final_poly, zmax);
            gp *= gammas[i];
        }
        j += k_whir;
    }
    if (acc != claim) throw VerifyError("WHIR FinalPolyConstraint");
}

// verifier/mod.rs:57-236
inline void verify(const MultiPk &mvk, const Proof &proof) {
    const SystemParams &params = mvk.params;
    const unsigned l_skip = params.l_skip;
    size_t num_airs = mvk.per_air.size();
    if (proof.trace_present.size() != num_airs || proof.public_values.size() != num_airs) throw VerifyError("num_airs");
    std::vector<size_t> ids;
    for (size_t a = 0; a < num_airs; a++) {
        if (proof.trace_present[a]) ids.push_back(a);
        else if (mvk.per_air[a].vk.is_required) throw VerifyError("RequiredAirNoVData");
    }
    if (ids.empty()) throw VerifyError("EmptyTraces");
    std::stable_sort(ids.begin(), ids.end(), [&](size_t a, size_t b) {
        if (proof.trace_vdata[a].log_height != proof.trace_vdata[b].log_height)
            return proof.trace_vdata[a].log_height > proof.trace_vdata[b].log_height;
        return a < b;
    });
    Fp omega_skip = two_adic_generator(l_skip);
    std::vector<Fp> omega_pows = powers(omega_skip, size_t(1) << l_skip);
    Transcript ts;
    ts.observe_commit(mvk.vk_pre_hash);
    ts.observe_commit(proof.common_main_commit);
    for (size_t a = 0; a < num_airs; a++) {
        const AirPk &pk = mvk.per_air[a];
        if (!pk.vk.is_required) ts.observe(Fp(proof.trace_present[a] ? 1 : 0));
        if (proof.trace_present[a]) {
            if (pk.preprocessed)
                ts.observe_commit(pk.preprocessed->commitment);
            else
                ts.observe(Fp((uint64_t)proof.trace_vdata[a].log_height));
            for (auto &c : proof.trace_vdata[a].cached_commitments) ts.observe_commit(c);
        }
        for (auto &pv : proof.public_values[a]) ts.observe(pv);
    }
    std::vector<int> n_per_trace;
    for (size_t a : ids) n_per_trace.push_back((int)proof.trace_vdata[a].log_height - (int)l_skip);
    std::vector<EF> r = verify_zerocheck_and_logup(ts, mvk, proof, ids, n_per_trace, omega_pows);
    // layouts (proof_shape.rs): common main, then one single-matrix layout per preprocessed/cached
    std::vector<StackedLayout> layouts;
    std::vector<std::vector<bool>> need_rot{{}};
    {
        std::vector<std::pair<size_t, unsigned>> meta;
        for (size_t a : ids) {
            meta.push_back({mvk.per_air[a].vk.common_main_width, (unsigned)proof.trace_vdata[a].log_height});
            need_rot[0].push_back(mvk.per_air[a].vk.need_rot);
        }
        layouts.emplace_back(l_skip, l_skip + params.n_stack, meta);
    }
    std::vector<Digest> commits{proof.common_main_commit};
    for (size_t a : ids) {
        const AirVk &vk = mvk.per_air[a].vk;
        unsigned lh = (unsigned)proof.trace_vdata[a].log_height;
        if (vk.has_preprocessed) {
            layouts.emplace_back(l_skip, l_skip + params.n_stack, std::vector<std::pair<size_t, unsigned>>{{vk.preprocessed_width, lh}});
            need_rot.push_back({vk.need_rot});
            commits.push_back(mvk.per_air[a].preprocessed->commitment);
        }
        for (size_t i = 0; i < vk.cached_main_widths.size(); i++) {
            layouts.emplace_back(l_skip, l_skip + params.n_stack, std::vector<std::pair<size_t, unsigned>>{{vk.cached_main_widths[i], lh}});
            need_rot.push_back({vk.need_rot});
            commits.push_back(proof.trace_vdata[a].cached_commitments[i]);
        }
    }
    std::vector<EF> u_prism = verify_stacked_reduction(ts, proof.stacking_proof, layouts, need_rot, l_skip, params.n_stack,
                                                       proof.batch_constraint_proof.column_openings, r);
    std::vector<EF> u_cube;
    EF cur = u_prism[0];
    for (unsigned i = 0; i < l_skip; i++) {
        u_cube.push_back(cur);
        cur = cur.square();
    }
    u_cube.insert(u_cube.end(), u_prism.begin() + 1, u_prism.end());
    verify_whir(ts, params, proof.whir_proof, proof.stacking_proof.stacking_openings, commits, u_cube);
}

inline void verify_encoded(const MultiPk &mvk, const uint8_t *bytes, size_t len) {
    Proof p = decode_proof(bytes, len);
    // decoding then re-encoding must reproduce the bytes
    std::vector<uint8_t> again = encode_proof(p);
    if (again.size() != len || memcmp(again.data(), bytes, len) != 0) throw VerifyError("re-encoding differs");
    verify(mvk, p);
}

}  // namespace oracle
