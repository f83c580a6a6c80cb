// ORACLE — test infrastructure only (see oracle/README.md).
//
// AIR description consumed by the prover: the symbolic constraint DAG, interactions, per-AIR
// verifying-key facts and the proving context. Restated from
//   crates/stark-backend/src/air_builders/symbolic/dag.rs:20-57        SymbolicExpressionNode / Dag
//   crates/stark-backend/src/air_builders/symbolic/symbolic_variable.rs:13-27   Entry
//   crates/stark-backend/src/interaction/mod.rs (Interaction<usize>: message, count, bus_index)
//   crates/stark-backend/src/keygen/types.rs:92-132                     StarkVerifyingParams / Key
//   crates/stark-backend/src/prover/types.rs:18-73                      proving key / context
//   crates/stark-backend/src/prover/logup_zerocheck/{single.rs,evaluator.rs}  evaluation order
#pragma once
#include "pcs.hpp"

namespace oracle {

enum NodeKind : uint32_t {
    N_VAR = 0,
    N_IS_FIRST = 1,
    N_IS_LAST = 2,
    N_IS_TRANSITION = 3,
    N_CONST = 4,
    N_ADD = 5,
    N_SUB = 6,
    N_NEG = 7,
    N_MUL = 8
};
enum EntryKind : uint32_t { E_PREPROCESSED = 0, E_MAIN = 1, E_PUBLIC = 2, E_CHALLENGE = 3 };

struct Node {
    NodeKind kind = N_CONST;
    // N_VAR
    EntryKind entry = E_MAIN;
    uint32_t part_index = 0, offset = 0, index = 0;
    // N_CONST
    Fp constant;
    // operators
    uint32_t left = 0, right = 0;
};

struct Interaction {
    std::vector<uint32_t> message;  // node indices
    uint32_t count = 0;             // node index
    uint32_t bus_index = 0;
    uint32_t count_weight = 0;
};

struct AirVk {
    std::vector<Node> nodes;
    std::vector<uint32_t> constraint_idx;
    std::vector<Interaction> interactions;
    bool need_rot = false;
    bool is_required = true;
    uint32_t max_constraint_degree = 0;
    uint32_t num_public_values = 0;
    bool has_preprocessed = false;
    size_t preprocessed_width = 0;
    std::vector<size_t> cached_main_widths;
    size_t common_main_width = 0;
};

struct CommittedTrace {
    Digest commitment;
    ColMat<Fp> trace;  // unstacked evaluations
    std::shared_ptr<StackedPcsData> data;
};

struct AirPk {
    AirVk vk;
    std::shared_ptr<CommittedTrace> preprocessed;  // null if none
};

struct SystemParams {
    unsigned l_skip = 0, n_stack = 0, w_stack = 0, log_blowup = 0;
    unsigned k_whir = 0;
    std::vector<size_t> whir_num_queries;  // per WHIR round
    unsigned mu_pow_bits = 0, query_phase_pow_bits = 0, folding_pow_bits = 0;
    unsigned logup_pow_bits = 0;
    unsigned max_constraint_degree = 0;
    size_t num_whir_rounds() const { return whir_num_queries.size(); }
};

struct MultiPk {
    std::vector<AirPk> per_air;
    SystemParams params;
    unsigned max_constraint_degree = 0;
    Digest vk_pre_hash;
};

struct AirCtx {
    size_t air_idx = 0;
    std::vector<std::shared_ptr<CommittedTrace>> cached_mains;
    ColMat<Fp> common_main;
    std::vector<Fp> public_values;
};

// One row view handed to the constraint evaluator: selectors, then per matrix (preprocessed,
// cached..., common) the local row and, if `need_rot`, the next row (single.rs:166-206).
template <class T>
struct RowView {
    T is_first, is_transition, is_last;
    const std::vector<std::vector<T>> *parts;  // as in `row_parts[1..]`
    bool needs_next, has_preprocessed;
    const std::vector<Fp> *public_values;

    const T &var(const Node &n) const {
        size_t mat;
        if (n.entry == E_PREPROCESSED) {
            mat = 0;
        } else {
            mat = (has_preprocessed ? 1 : 0) + n.part_index;
        }
        size_t idx = needs_next ? 2 * mat + n.offset : mat;
        assert(needs_next || n.offset == 0);
        return (*parts)[idx][n.index];
    }
};

inline void lift(Fp &dst, Fp c) { dst = c; }
inline void lift(EF &dst, Fp c) { dst = EF(c); }

// evaluator.rs:53-96 + symbolic_expression.rs eval: values of every DAG node, in order.
template <class T>
inline std::vector<T> eval_nodes(const AirVk &vk, const RowView<T> &row) {
    std::vector<T> v(vk.nodes.size());
    for (size_t i = 0; i < vk.nodes.size(); i++) {
        const Node &n = vk.nodes[i];
        switch (n.kind) {
            case N_VAR:
                if (n.entry == E_PUBLIC)
                    lift(v[i], (*row.public_values)[n.index]);
                else if (n.entry == E_CHALLENGE)
                    throw std::runtime_error("ChallengeNotSupported");
                else
                    v[i] = row.var(n);
                break;
            case N_IS_FIRST: v[i] = row.is_first; break;
            case N_IS_LAST: v[i] = row.is_last; break;
            case N_IS_TRANSITION: v[i] = row.is_transition; break;
            case N_CONST: lift(v[i], n.constant); break;
            case N_ADD: v[i] = v[n.left] + v[n.right]; break;
            case N_SUB: v[i] = v[n.left] - v[n.right]; break;
            case N_NEG: v[i] = -v[n.left]; break;
            case N_MUL: v[i] = v[n.left] * v[n.right]; break;
        }
    }
    return v;
}

// single.rs:82-94 : sum_i lambda^i * constraint_i
template <class T>
inline EF acc_constraints(const AirVk &vk, const RowView<T> &row, const std::vector<EF> &lambda_pows) {
    std::vector<T> nodes = eval_nodes(vk, row);
    EF acc = EF::zero();
    for (size_t i = 0; i < vk.constraint_idx.size(); i++)
        acc += lambda_pows[i] * nodes[vk.constraint_idx[i]];
    return acc;
}

// single.rs:127-153 : per interaction (numer = count, denom = beta^len * (bus+1) + sum beta^j msg_j)
template <class T>
inline std::vector<std::pair<T, EF>> eval_interactions(const AirVk &vk, const RowView<T> &row,
                                                       const std::vector<EF> &beta_pows) {
    std::vector<T> nodes = eval_nodes(vk, row);
    std::vector<std::pair<T, EF>> out;
    for (const Interaction &it : vk.interactions) {
        Fp b = Fp((uint64_t)it.bus_index + 1);
        size_t len = it.message.size();
        EF denom = beta_pows[len] * b;
        for (size_t j = 0; j < len; j++) denom += beta_pows[j] * nodes[it.message[j]];
        out.push_back({nodes[it.count], denom});
    }
    return out;
}

// single.rs:100-125
template <class T>
inline std::array<EF, 2> acc_interactions(const AirVk &vk, const RowView<T> &row,
                                          const std::vector<EF> &beta_pows,
                                          const std::vector<EF> &eq_3bs) {
    auto evals = eval_interactions(vk, row, beta_pows);
    EF numer = EF::zero(), denom = EF::zero();
    for (size_t i = 0; i < evals.size(); i++) {
        numer += eq_3bs[i] * evals[i].first;
        denom += eq_3bs[i] * evals[i].second;
    }
    return {numer, denom};
}

}  // namespace oracle
