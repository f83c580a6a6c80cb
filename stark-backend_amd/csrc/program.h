// Constraint / interaction programs: the symbolic DAG of one AIR flattened (on the host, once per
// proof) into a linear register program that one GPU lane interprets per evaluation point.
//
// Replaces (reference): the rule compiler + interpreter of the CUDA backend --
// crates/cuda-backend/src/logup_zerocheck/rules/ (host), cuda/include/{codec,dag_entry,eval_ctx}.cuh
// (device) -- and, for semantics, crates/stark-backend/src/prover/logup_zerocheck/single.rs:82-153.
//
// A program evaluates only the DAG nodes its roots need. Intermediate values live in `slots`
// assigned by liveness (a value's slot is recycled after its last use), so the per-lane scratch
// footprint is the DAG's maximum live set, not its size. Leaves (trace cells, constants, public
// values, selectors) are operands, never slots. Accumulation is part of the program:
//   ACC  acc, w, x :  out[acc] += W[w] * x        (W: extension-field weights, host supplied)
//   DEN  w, x      :  den += W[w] * x             (LogUp-GKR leaf mode)
//   LEAF k, x      :  emit leaf k = (x, den + C[k]); den = 0
#pragma once
#include <iterator>
#include <vector>

#include "host.h"

namespace ovm {

enum Op : uint8_t { OP_ADD = 0, OP_SUB, OP_MUL, OP_NEG, OP_ACC, OP_DEN, OP_LEAF };
enum OperandKind : uint8_t { K_SLOT = 0, K_COL, K_CONST };

struct Instr {
    uint8_t op, ak, bk, acc;
    uint16_t dst, fwd;  // fwd: FWD_* hints (an executor may ignore them: they never change what is computed)
    uint32_t a, b;  // operand payloads: slot / flat column / Montgomery constant / weight index
};
// Forwarding hints: operand a / b (a slot) is the result of the instruction right before this one, and -- NOSTORE, set on
// that earlier instruction -- nothing but the next instruction ever reads its result. An executor that keeps the last
// result in registers skips the slot read, and for NOSTORE the slot write as well (stages.hip:
// mle_round_points_lds_kernel, where slot traffic is LDS bandwidth).
enum : uint16_t { FWD_A = 1, FWD_B = 2, FWD_NOSTORE = 4 };
static_assert(sizeof(Instr) == 16, "Instr is 16 bytes");

struct Program {
    std::vector<Instr> code;
    uint32_t num_slots = 0;
    uint32_t num_weights = 0;
    std::vector<uint32_t> used_cols;  // flat column indices the program reads (sorted)
};

// Flat column numbering of one trace's view: 0,1,2 = is_first, is_transition, is_last, then for
// every matrix of view_mats (preprocessed, cached..., common) its columns, local then (if the AIR
// needs rotation) next.
// Flat column numbering of one trace view, in the reference's row_parts order
// (logup_zerocheck/single.rs:147-158): selectors 0..2, then per matrix [local cols, rotated cols]
// for the preprocessed trace (if any), the cached main partitions, the common main.
struct ColumnMap {
    bool need_rot = false;
    size_t pre_width = 0;            // preprocessed trace width (0 = none)
    std::vector<size_t> part_width;  // widths of [cached..., common]
    size_t total = 3;
    size_t pre_base = 3;
    std::vector<size_t> part_base;  // flat index of part p, offset 0
    void finish() {
        part_base.clear();
        size_t cur = 3;
        pre_base = cur;
        cur += pre_width * (need_rot ? 2 : 1);
        for (size_t w : part_width) {
            part_base.push_back(cur);
            cur += w * (need_rot ? 2 : 1);
        }
        total = cur;
    }
    uint32_t col(uint32_t part, uint32_t offset, uint32_t index) const {
        return (uint32_t)(part_base[part] + (offset ? part_width[part] : 0) + index);
    }
    uint32_t pre_col(uint32_t offset, uint32_t index) const {
        return (uint32_t)(pre_base + (offset ? pre_width : 0) + index);
    }
};

static constexpr uint32_t ROOT_ONE_NODE = 0xFFFFFFFFu;  // Root::node: the constant one (a term of linear_terms)
struct Root {
    uint32_t node;
    uint8_t op;      // OP_ACC / OP_DEN / OP_LEAF
    uint8_t acc;     // accumulator for OP_ACC
    uint32_t weight; // weight index (ACC, DEN) or leaf index (LEAF)
};

inline ColumnMap column_map_of(const AirKey &air) {
    ColumnMap cm;
    cm.need_rot = air.need_rot;
    cm.pre_width = air.preprocessed_width;
    cm.part_width = air.cached_widths;
    cm.part_width.push_back(air.common_main_width);
    cm.finish();
    return cm;
}

// Roots of the three programs of an AIR. The weight indices they carry are positions in arrays
// the prover fills per proof (lambda powers; eq_3b * beta powers; beta powers), so the program
// STRUCTURE depends on the AIR only.
inline std::vector<Root> roots_zerocheck(const AirKey &a) {
    std::vector<Root> roots;
    for (size_t i = 0; i < a.constraint_idx.size(); i++) roots.push_back({a.constraint_idx[i], OP_ACC, 0, (uint32_t)i});
    return roots;
}
// logup round sums: accumulator 1 += w * message field, accumulator 0 += w * count; weights in emission order
inline std::vector<Root> roots_logup(const AirKey &a, uint32_t *num_weights = nullptr) {
    std::vector<Root> roots;
    uint32_t w = 0;
    for (auto &it : a.interactions) {
        for (size_t j = 0; j < it.message.size(); j++) roots.push_back({it.message[j], OP_ACC, 1, w++});
        roots.push_back({it.count, OP_ACC, 0, w++});
    }
    if (num_weights) *num_weights = w;
    return roots;
}
// GKR leaves: per interaction the message fields feed the denominator (weight = beta^j), the count closes leaf k
inline std::vector<Root> roots_leaves(const AirKey &a) {
    std::vector<Root> roots;
    for (size_t k = 0; k < a.interactions.size(); k++) {
        const Interaction &it = a.interactions[k];
        for (size_t j = 0; j < it.message.size(); j++) roots.push_back({it.message[j], OP_DEN, 0, (uint32_t)j});
        roots.push_back({it.count, OP_LEAF, 0, (uint32_t)k});
    }
    return roots;
}

// Compiles the nodes the `roots` need into a register program. Roots fire strictly in the given order (a LEAF closes
// the DENs before it). Public values are folded to constants.
//
// Scheduling is DEMAND-DRIVEN: for each root in turn its expression is emitted depth-first (the operand that needs more
// registers first, Sethi-Ullman), so a value is computed right before its first use and its slot is recycled after the
// last one -- the slot count is the live set of a depth-first walk, not of the DAG's construction order. A node whose whole
// expression tree costs at most `remat_limit` instructions is never kept: every use recomputes it (limit 1 = an operation
// on two leaves, e.g. the `q + q` of x xor q = x + q - x (q + q), which p3-keccak-air's constraints share between columns
// that are hundreds of constraints apart). compile_program() raises the limit until the program fits `max_slots`:
// the first version emitted nodes in DAG index order and the real Keccak-f AIR (39 K instructions) needed 645 slots --
// more than any executor holds -- where this order needs a dozen.
struct CompileStats {
    uint32_t remat_limit = 0, recomputed = 0;
};
inline Program compile_program_with(const AirKey &air, const ColumnMap &cm, const std::vector<Fp> &public_values,
                                    const std::vector<Root> &roots, uint32_t num_weights, uint32_t remat_limit,
                                    CompileStats *stats = nullptr) {
    size_t n = 0;
    for (auto &r : roots)
        if (r.node != ROOT_ONE_NODE) n = std::max(n, (size_t)r.node + 1);
    if (n > air.nodes.size()) throw OvmFail(1, "program compiler: root out of range");
    auto is_op = [&](uint32_t i) { return air.nodes[i].kind >= N_ADD; };
    auto binary = [&](uint32_t i) { return air.nodes[i].kind != N_NEG; };
    // Reachable operation nodes in index order (children precede their parents): tree cost, register need, and -- for
    // the nodes that are kept -- how many kept parents / root firings will read them.
    std::vector<char> need(n, 0);
    {
        std::vector<uint32_t> stack;
        for (auto &r : roots)
            if (r.node != ROOT_ONE_NODE && !need[r.node]) need[r.node] = 1, stack.push_back(r.node);
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (!is_op(i)) continue;
            const Node &nd = air.nodes[i];
            if (nd.left >= i || (binary(i) && nd.right >= i)) throw OvmFail(1, "program compiler: DAG not topological");
            if (!need[nd.left]) need[nd.left] = 1, stack.push_back(nd.left);
            if (binary(i) && !need[nd.right]) need[nd.right] = 1, stack.push_back(nd.right);
        }
    }
    std::vector<uint32_t> cost(n, 0), regs(n, 0), uses(n, 0);
    size_t lo = n;
    for (size_t i = 0; i < n; i++) {
        if (!need[i] || !is_op((uint32_t)i)) continue;
        lo = std::min(lo, i);
        const Node &nd = air.nodes[i];
        const uint64_t c = 1ull + cost[nd.left] + (binary((uint32_t)i) ? cost[nd.right] : 0);
        cost[i] = (uint32_t)std::min<uint64_t>(c, 1u << 30);
        const uint32_t rl = regs[nd.left], rr = binary((uint32_t)i) ? regs[nd.right] : 0;
        regs[i] = std::max<uint32_t>(1, rl == rr ? rl + (rl ? 1 : 0) : std::max(rl, rr));
    }
    auto remat = [&](uint32_t i) { return cost[i] <= remat_limit; };
    for (size_t i = lo; i < n; i++) {
        if (!need[i] || !is_op((uint32_t)i) || remat((uint32_t)i)) continue;  // (a recomputed node's operands are recomputed too)
        const Node &nd = air.nodes[i];
        if (is_op(nd.left)) uses[nd.left]++;
        if (binary((uint32_t)i) && is_op(nd.right)) uses[nd.right]++;
    }
    for (auto &r : roots)
        if (r.node != ROOT_ONE_NODE && is_op(r.node)) uses[r.node]++;

    Program prog;
    prog.num_weights = num_weights;
    std::vector<int32_t> slot_of(n, -1);
    std::vector<uint32_t> free_slots;
    std::vector<char> col_used(cm.total, 0);
    struct Operand {
        uint8_t kind;
        uint32_t payload;
        int32_t node;  // the operation node whose slot this is (-1: a leaf)
    };
    auto leaf_operand = [&](uint32_t node) -> Operand {
        const Node &nd = air.nodes[node];
        switch (nd.kind) {
            case N_VAR:
                if (nd.entry == E_PUBLIC) return {K_CONST, public_values.at(nd.index).v, -1};
                if (nd.entry == E_MAIN) {
                    if (nd.part_index >= cm.part_width.size() || nd.index >= cm.part_width[nd.part_index] ||
                        (nd.offset && !cm.need_rot))
                        throw OvmFail(1, "main variable out of range");
                    const uint32_t c = cm.col(nd.part_index, nd.offset, nd.index);
                    col_used[c] = 1;
                    return {K_COL, c, -1};
                }
                if (nd.entry == E_PREPROCESSED) {
                    if (nd.index >= cm.pre_width || (nd.offset && !cm.need_rot))
                        throw OvmFail(1, "preprocessed variable out of range");
                    const uint32_t c = cm.pre_col(nd.offset, nd.index);
                    col_used[c] = 1;
                    return {K_COL, c, -1};
                }
                throw OvmFail(801, "challenge variables not supported");
            case N_IS_FIRST: col_used[0] = 1; return {K_COL, 0, -1};
            case N_IS_TRANSITION: col_used[1] = 1; return {K_COL, 1, -1};
            case N_IS_LAST: col_used[2] = 1; return {K_COL, 2, -1};
            case N_CONST: return {K_CONST, nd.constant.v, -1};
            default: throw OvmFail(1, "program compiler: not a leaf");
        }
    };
    auto take_slot = [&]() {
        uint32_t s;
        if (!free_slots.empty()) {
            s = free_slots.back();
            free_slots.pop_back();
        } else {
            s = prog.num_slots++;
        }
        if (s > 0xFFFF) throw OvmFail(801, "program needs too many slots");
        return s;
    };
    // one read of an operand by an instruction or a root firing is over: recycle its slot when nothing else will read it
    auto consumed = [&](const Operand &o) {
        if (o.node < 0) return;
        if (remat((uint32_t)o.node)) {
            free_slots.push_back(o.payload);
        } else if (--uses[o.node] == 0) {
            free_slots.push_back((uint32_t)slot_of[o.node]);
            slot_of[o.node] = -2;
        }
    };
    // forwarding hints (see FWD_*): `in` is about to be appended and reads operands l (and r)
    auto mark_forwards = [&](Instr &in, const Operand &l, const Operand *r) {
        if (prog.code.empty()) return;
        Instr &prev = prog.code.back();
        if (prev.op > OP_NEG) return;  // (only arithmetic instructions produce a slot value)
        const bool fa = l.kind == K_SLOT && l.node >= 0 && l.payload == prev.dst;
        const bool fb = r && r->kind == K_SLOT && r->node >= 0 && r->payload == prev.dst;
        if (!fa && !fb) return;
        if (fa) in.fwd |= FWD_A;
        if (fb) in.fwd |= FWD_B;
        // the value dies here when these reads are all that is left of it (a recomputed node's instance has one reader)
        const Operand &o = fa ? l : *r;
        const uint32_t reads = (fa ? 1u : 0u) + (fb ? 1u : 0u);
        if (remat((uint32_t)o.node) || uses[o.node] == reads) prev.fwd |= FWD_NOSTORE;
    };
    // value of `node` as an operand, emitting whatever it takes (explicit stack: an expression can be thousands deep)
    struct Frame {
        uint32_t node;
        uint8_t state;  // 0: nothing done, 1: first operand done, 2: both done
        bool right_first;
        Operand first, second;
    };
    std::vector<Frame> stack;
    auto value_of = [&](uint32_t root_node) -> Operand {
        if (!is_op(root_node)) return leaf_operand(root_node);
        if (!remat(root_node) && slot_of[root_node] >= 0) return {K_SLOT, (uint32_t)slot_of[root_node], (int32_t)root_node};
        Operand result{};
        stack.clear();
        stack.push_back({root_node, 0, false, {}, {}});
        while (!stack.empty()) {
            Frame &f = stack.back();
            const Node &nd = air.nodes[f.node];
            const bool bin = binary(f.node);
            if (f.state == 0) f.right_first = bin && regs[nd.right] > regs[nd.left];
            if (f.state < (bin ? 2 : 1)) {
                const uint32_t child = (f.state == 0) == f.right_first && bin ? nd.right : nd.left;
                Operand *into = f.state == 0 ? &f.first : &f.second;
                f.state++;
                if (!is_op(child)) {
                    *into = leaf_operand(child);
                } else if (!remat(child) && slot_of[child] >= 0) {
                    *into = {K_SLOT, (uint32_t)slot_of[child], (int32_t)child};
                } else if (!remat(child) && slot_of[child] == -2) {
                    throw OvmFail(1, "program compiler: value not live");
                } else {
                    stack.push_back({child, 0, false, {}, {}});  // (invalidates f: the loop re-reads stack.back())
                }
                continue;
            }
            const Operand &l = bin && f.right_first ? f.second : f.first;
            const Operand &r = bin && f.right_first ? f.first : f.second;
            Instr in{};
            in.op = nd.kind == N_ADD ? OP_ADD : nd.kind == N_SUB ? OP_SUB : nd.kind == N_MUL ? OP_MUL : OP_NEG;
            in.ak = l.kind, in.a = l.payload;
            if (bin) in.bk = r.kind, in.b = r.payload;
            const uint32_t s = take_slot();  // (before the operands' slots are recycled: dst never aliases an operand)
            in.dst = (uint16_t)s;
            mark_forwards(in, l, bin ? &r : nullptr);
            prog.code.push_back(in);
            consumed(l);
            if (bin) consumed(r);
            const uint32_t me = f.node;
            if (remat(me)) {
                if (stats && slot_of[me] == -3) stats->recomputed++;
                slot_of[me] = -3;  // (only marks "was computed before" for the statistics)
            } else {
                slot_of[me] = (int32_t)s;
            }
            const Operand mine{K_SLOT, s, (int32_t)me};
            stack.pop_back();
            if (stack.empty()) {
                result = mine;
            } else {
                Frame &p = stack.back();
                (p.state == 1 ? p.first : p.second) = mine;
            }
        }
        return result;
    };
    for (const Root &r : roots) {
        const Operand o = r.node == ROOT_ONE_NODE ? Operand{K_CONST, Fp::one().v, -1} : value_of(r.node);
        Instr in{};
        in.op = r.op;
        in.acc = r.acc;
        in.ak = o.kind, in.a = o.payload;
        in.b = r.weight;
        mark_forwards(in, o, nullptr);
        prog.code.push_back(in);
        consumed(o);
    }
    for (size_t c = 0; c < cm.total; c++)
        if (col_used[c]) prog.used_cols.push_back((uint32_t)c);
    if (prog.num_slots == 0) prog.num_slots = 1;
    if (stats) stats->remat_limit = remat_limit;
    return prog;
}

// The program with the smallest recomputation limit that fits `max_slots` (the plain interpreter holds up to 512 slots
// in private memory, the staged VM ~50 rows of LDS). Limits 1, 2, 4, ...: in the end every shared subexpression is
// recomputed at each use and the slot count is the register need of one expression tree.
static constexpr uint32_t PROGRAM_SLOT_TARGET = 40;
inline Program compile_program(const AirKey &air, const ColumnMap &cm, const std::vector<Fp> &public_values,
                               const std::vector<Root> &roots, uint32_t num_weights, uint32_t max_slots = PROGRAM_SLOT_TARGET,
                               CompileStats *stats = nullptr) {
    Program best = compile_program_with(air, cm, public_values, roots, num_weights, 0, stats);
    // (a program that fits the target recomputes nothing; otherwise trade instructions for slots while that at most doubles
    // the code, and past the hard limit of the executors at any price)
    for (uint32_t limit = 1; best.num_slots > max_slots && limit <= (1u << 20); limit *= 2) {  // 1, 2, 4, ...
        CompileStats st;
        Program p = compile_program_with(air, cm, public_values, roots, num_weights, limit, &st);
        const bool must = best.num_slots > 512;
        if (p.num_slots < best.num_slots && (must || p.code.size() <= 2 * best.code.size())) {
            best = std::move(p);
            if (stats) *stats = st;
        } else if (!must && p.code.size() > 2 * best.code.size()) {
            break;
        }
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// Words of the cached-column interpreter (cached_vm.cuh). The plain interpreter further down decodes operand kinds per
// instruction and loads trace cells where they are used; on gfx950 that is bound by scalar / branch issue and by the serial
// decode -> load -> use chain, not by arithmetic. The staged form moves every decision to the host, once per proof: every
// operand is an LDS row, the result register or an immediate (constants on the left are commuted, folded or
// materialised), operand fields are pre-multiplied word offsets, the operation is one bit per kind.
// ---------------------------------------------------------------------------------------------
enum StagedOp : uint32_t {
    S_MUL = 1, S_ADD = 2, S_SUB = 4, S_RSUB = 8 /* b - a */, S_MOV = 32 /* dst = a */, S_MOVC = 64 /* dst = b */
};
static constexpr uint32_t S_B_CONST = 0x100;
// Forwarding (the compiler's FWD_* hints): operand a / b is the result of the word right before this one and comes from the
// result register, not from an LDS row; S_NOSTORE: the result is read by the next word only, through that register.
static constexpr uint32_t S_FWD_A = 0x200, S_FWD_B = 0x400, S_NOSTORE = 0x800;

struct StagedInstr {
    uint32_t op;   // StagedOp | flags
    uint32_t dst;  // LDS row
    uint32_t a;    // LDS row
    uint32_t b;    // LDS row / Montgomery constant
};

// ---------------------------------------------------------------------------------------------
// Cached-column form of a program (round 0's accumulation programs and the LogUp-GKR leaf programs; cached_vm.cuh).
//
// The block-staged VM of rounds 2-4 staged every column a block read, every block. Here the LDS image is ONE pool of rows shared by
// a CACHE of columns that survives from block to block and by the program's live values (a slot needs a row only from
// where its value is computed to its last read: the real Keccak-f program has 8 slots and, nine instructions in ten, at
// most three live values). A block's load list names only the columns that are not cached (column -> LDS row), chosen
// against the program's own future -- the column evicted is the one whose next read is farthest away.
// The loads of block b + 1 are requested when block b starts, into REGISTERS, and written to their rows when block b + 1
// starts: nothing in LDS is reserved for data in flight.
// The program is ONE stream of 16-byte words (the kernel fetches them four at a time, one group ahead):
//   arithmetic   {StagedOp | flags, dst row, a row, b row | constant}: the first operand is the RESULT REGISTER (the value
//                of the word before: S_FWD_A) or row a, the second the constant b (S_B_CONST), the result register
//                (S_FWD_B, only beside S_FWD_A) or row b; S_NOSTORE: the result stays in the result register only.
//                A value forwarded as the SECOND operand of the compiled instruction is commuted here (b - result
//                becomes S_RSUB), a negation is 0 - x: the kernel has no S_NEG
//   C_ACC        {C_ACC | w0, acc << 31 | w1, w2, w3}     accumulator acc += W * (result register); the weight's
//                Montgomery words are below 2^31, bit 31 of the first two is free
//   C_BLOCK      {C_BLOCK | n_land | n_issue << 8, record, 0, 0}   write the columns requested by the block word before
//                to their rows, request the next block's; a record is MAX_LOADS row offsets (of the columns to write)
//                followed by MAX_LOADS flat columns (to request). Always the first word of a group of four (the
//                group before is padded with no-ops)
// An accumulation reads the value it weighs from the result register: what it weighs is computed right before it, or
// brought there by a move (a column, a slot, a constant).
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t C_ACC = 0x80000000u, C_BLOCK = 0x40000000u;
// DEN / LEAF programs (the LogUp-GKR leaves): a denominator term is an accumulation into accumulator 0; a LEAF brings its
// numerator into the result register and is followed by a marker word {C_BLOCK | C_LEAF, interaction, 0, 0} in the LAST
// position of its group of four (the asm skips it like any block-kind word that is not first; the kernel's C++ loop sees it
// after the group, emits (numerator, accumulator 0 + constant) and clears the accumulator).
static constexpr uint32_t C_LEAF = 0x20000000u;
struct CachedLayout {
    uint32_t rows = 12;      // LDS rows of the image: shared by the cached columns and the program's live values
    uint32_t max_loads = 8;  // columns a block may bring (<= CachedProgram::MAX_LOADS: the kernel's landing registers)
};
struct CachedProgram {
    static constexpr uint32_t MAX_LOADS = 8;
    std::vector<StagedInstr> words;  // a multiple of 8 (padded with no-ops) + one more group of four
    std::vector<uint32_t> loads;     // one record per block word: MAX_LOADS LDS rows to write, MAX_LOADS flat columns to request
    uint32_t num_rows = 0, n_blocks = 0, n_leaves = 0;
    uint64_t n_loads = 0;  // columns brought per evaluation (statistics)
    // (word, weight index) of every accumulation word: the weights are inlined in the words, and a program staged before
    // its weights exist (on the compile thread, while the device commits) gets them patched in (patch_weights)
    std::vector<std::pair<uint32_t, uint32_t>> weight_at;
    void patch_weights(const std::vector<FpExt> &weights) {
        for (const auto &wa : weight_at) {
            if (wa.second >= weights.size()) throw OvmFail(1, "patch_weights: weight index out of range");
            const FpExt &w = weights[wa.second];
            StagedInstr &in = words[wa.first];
            in.op = C_ACC | w.c[0].v;
            in.dst = (in.dst & 0x80000000u) | w.c[1].v;
            in.a = w.c[2].v;
            in.b = w.c[3].v;
        }
    }
    // operand rows -> word offsets for `rows_per_lane` trace rows per lane
    void scale_rows(uint32_t rows_per_lane) {
        const uint32_t k = rows_per_lane * 256;
        for (auto &in : words) {
            if (in.op & (C_ACC | C_BLOCK)) continue;
            in.dst *= k;
            in.a *= k;
            if (!(in.op & S_B_CONST)) in.b *= k;
        }
        for (size_t i = 0; i < loads.size(); i++)
            if (i % (2 * MAX_LOADS) < MAX_LOADS) loads[i] *= k;
    }
};

// What stage_cached decides per instruction before it lays the rows out: is the result written to LDS at all (a value
// read only by the next instruction stays in the result register), and until which instruction is it read from there.
struct CachedLiveness {
    std::vector<char> stored;        // arithmetic instruction i writes its result to a row
    std::vector<int32_t> last_read;  // ... which is read from LDS for the last time by this instruction (< i: never)
    std::vector<uint32_t> live;      // stored values alive while instruction i runs (its own result included)
    uint32_t min_rows = 2;           // rows the program cannot do without: max over i of live[i] + columns read by i
};
inline CachedLiveness cached_liveness(const Program &p) {
    const size_t n = p.code.size();
    CachedLiveness lv;
    lv.stored.assign(n, 0);
    lv.last_read.assign(n, -1);
    lv.live.assign(n, 0);
    std::vector<int32_t> def(p.num_slots + 1, -1);  // the instruction whose result a slot holds
    auto is_binary = [](const Instr &in) { return in.op == OP_ADD || in.op == OP_SUB || in.op == OP_MUL; };
    for (size_t i = 0; i < n; i++) {
        const Instr &in = p.code[i];
        if (in.op > OP_LEAF) throw OvmFail(1, "stage_cached: unknown operation");
        // reads from LDS: slot operands that are not forwarded from the instruction before
        if (in.ak == K_SLOT && !(in.fwd & FWD_A) && def.at(in.a) >= 0) lv.last_read[def[in.a]] = (int32_t)i;
        if (is_binary(in) && in.bk == K_SLOT && !(in.fwd & FWD_B) && def.at(in.b) >= 0) lv.last_read[def[in.b]] = (int32_t)i;
        if (in.op <= OP_NEG) {
            const bool next_takes_it = i + 1 < n && (p.code[i + 1].fwd & (FWD_A | FWD_B));
            lv.stored[i] = !((in.fwd & FWD_NOSTORE) && next_takes_it);
            def.at(in.dst) = (int32_t)i;
        }
    }
    std::vector<int32_t> delta(n + 1, 0);
    for (size_t i = 0; i < n; i++)
        if (lv.stored[i]) {
            delta[i]++;
            delta[std::max<size_t>(i, lv.last_read[i] < 0 ? 0 : (size_t)lv.last_read[i]) + 1]--;
        }
    int32_t cur = 0;
    for (size_t i = 0; i < n; i++) {
        cur += delta[i];
        lv.live[i] = (uint32_t)cur;
        const Instr &in = p.code[i];
        uint32_t cols = in.ak == K_COL ? 1 : 0;
        if (is_binary(in) && in.bk == K_COL && !(in.ak == K_COL && in.a == in.b)) cols++;
        lv.min_rows = std::max(lv.min_rows, lv.live[i] + cols);
    }
    return lv;
}

// (defer_weights: `weights` is not looked at; every accumulation word carries a zero weight until patch_weights)
inline CachedProgram stage_cached(const Program &p, const std::vector<FpExt> &weights, const CachedLayout &layout = CachedLayout(),
                                  bool defer_weights = false) {
    const CachedLiveness lv = cached_liveness(p);
    const uint32_t T = layout.rows, L = std::min(layout.max_loads, CachedProgram::MAX_LOADS);
    if (T < lv.min_rows || L < 2) throw OvmFail(1, "stage_cached: the image is too small for this program");
    CachedProgram cp;
    cp.num_rows = T;
    const size_t n = p.code.size();
    auto is_binary = [](const Instr &in) { return in.op == OP_ADD || in.op == OP_SUB || in.op == OP_MUL; };
    // the columns instruction i reads, and per column the instructions that read it (for the eviction choice)
    auto cols_of = [&](const Instr &in, uint32_t (&c)[2]) {
        int k = 0;
        if (in.ak == K_COL) c[k++] = in.a;
        if (is_binary(in) && in.bk == K_COL && !(k && c[0] == in.b)) c[k++] = in.b;
        return k;
    };
    uint32_t max_col = 0;
    for (const Instr &in : p.code) {
        uint32_t c[2];
        for (int k = cols_of(in, c); k-- > 0;) max_col = std::max(max_col, c[k]);
    }
    std::vector<std::vector<uint32_t>> readers(max_col + 1);
    for (size_t i = 0; i < n; i++) {
        uint32_t c[2];
        for (int k = cols_of(p.code[i], c), j = 0; j < k; j++) readers[c[j]].push_back((uint32_t)i);
    }
    auto next_read = [&](uint32_t col, size_t from) -> size_t {  // first reader at or after `from` (n: none)
        const auto &r = readers[col];
        auto it = std::lower_bound(r.begin(), r.end(), (uint32_t)from);
        return it == r.end() ? n : *it;
    };
    // ---- rows: ONE pool for cached columns and live values ----
    // A block is a run of instructions with  columns read + most values alive at once <= T  and at most L columns to
    // bring. Its columns keep (or get, at its start) their rows for the whole block; a value takes, where it is computed, a
    // row that holds neither a column of the block nor a live value: a free one, or the cached column read again last.
    static constexpr uint32_t FREE = ~0u, VALUE = ~0u - 1;
    std::vector<int32_t> row_of(max_col + 1, -1);  // LDS row of a cached column
    std::vector<uint32_t> holds(T, FREE);          // a column, VALUE, or FREE
    std::vector<int32_t> value_row(n, -1);         // row of the result of instruction i while it is alive
    std::vector<std::vector<uint32_t>> dying(n + 1);  // values whose last read is instruction i
    struct Block {
        size_t first, last;                                // instructions [first, last)
        std::vector<std::pair<uint32_t, uint32_t>> loads;  // (column, row) to bring
        std::vector<std::pair<uint32_t, uint32_t>> rows;   // (column, row) of every column the block reads
    };
    std::vector<Block> blocks;
    std::vector<uint32_t> stamp(max_col + 1, 0);
    for (size_t first = 0; first < n;) {
        const uint32_t id = (uint32_t)blocks.size() + 1;
        std::vector<uint32_t> used, fresh;
        uint32_t peak = 0;
        size_t last = first;
        for (; last < n; last++) {
            uint32_t c[2];
            const int k = cols_of(p.code[last], c);
            uint32_t add_used = 0, add_fresh = 0;
            for (int j = 0; j < k; j++)
                if (stamp[c[j]] != id) add_used++, add_fresh += row_of[c[j]] < 0;
            if (used.size() + add_used + std::max(peak, lv.live[last]) > T || fresh.size() + add_fresh > L) break;
            peak = std::max(peak, lv.live[last]);
            for (int j = 0; j < k; j++)
                if (stamp[c[j]] != id) {
                    stamp[c[j]] = id;
                    used.push_back(c[j]);
                    if (row_of[c[j]] < 0) fresh.push_back(c[j]);
                }
        }
        if (last == first) throw OvmFail(1, "stage_cached: an instruction does not fit a block");
        // the row a newcomer takes: a free one, else the cached column outside this block that is read again last
        auto take_row = [&](size_t from) -> uint32_t {
            uint32_t best = FREE;
            size_t best_next = 0;
            for (uint32_t r = 0; r < T; r++) {
                if (holds[r] == FREE) return r;
                if (holds[r] == VALUE || stamp[holds[r]] == id) continue;
                const size_t nx = next_read(holds[r], from) + 1;
                if (nx > best_next) best = r, best_next = nx;
            }
            if (best == FREE) throw OvmFail(1, "stage_cached: internal (no row left)");
            row_of[holds[best]] = -1;
            return best;
        };
        Block blk{first, last, {}, {}};
        for (uint32_t col : fresh) {
            const uint32_t r = take_row(last);
            holds[r] = col;
            row_of[col] = (int32_t)r;
            blk.loads.push_back({col, r});
        }
        for (uint32_t c : used) blk.rows.push_back({c, (uint32_t)row_of[c]});
        // the values of the block: rows taken where they are computed, given back after their last read
        for (size_t i = first; i < last; i++) {
            if (lv.stored[i]) {
                const uint32_t r = take_row(last);
                holds[r] = VALUE;
                value_row[i] = (int32_t)r;
                dying[std::max<size_t>(i, lv.last_read[i] < 0 ? 0 : (size_t)lv.last_read[i])].push_back((uint32_t)i);
            }
            for (uint32_t v : dying[i]) holds[(size_t)value_row[v]] = FREE;
        }
        cp.n_loads += fresh.size();
        blocks.push_back(std::move(blk));
        first = last;
    }
    cp.n_blocks = (uint32_t)blocks.size();
    // ---- words ----
    // (a no-op keeps the result register: a forwarded move that stores nothing)
    const StagedInstr nop{S_MOV | S_B_CONST | S_FWD_A | S_NOSTORE, 0, 0, 0};
    const size_t npos = ~size_t(0);
    auto block_word = [&](size_t land, size_t issue) {  // write the loads of block `land`, request those of block `issue` (npos: none)
        while (cp.words.size() % 4) cp.words.push_back(nop);  // a block word is the first word of a group of four
        const uint32_t nl = land < blocks.size() ? (uint32_t)blocks[land].loads.size() : 0;
        const uint32_t ni = issue < blocks.size() ? (uint32_t)blocks[issue].loads.size() : 0;
        const uint32_t rec = (uint32_t)(cp.loads.size() / (2 * CachedProgram::MAX_LOADS));
        for (uint32_t j = 0; j < CachedProgram::MAX_LOADS; j++) cp.loads.push_back(j < nl ? blocks[land].loads[j].second : 0);
        for (uint32_t j = 0; j < CachedProgram::MAX_LOADS; j++) cp.loads.push_back(j < ni ? blocks[issue].loads[j].first : 0);
        cp.words.push_back(StagedInstr{C_BLOCK | nl | (ni << 8), rec, 0, 0});
    };
    if (!blocks.empty()) block_word(npos, 0);
    std::vector<int32_t> cur_row(max_col + 1, -1);
    std::vector<int32_t> slot_value(p.num_slots + 1, -1);  // the instruction whose result a slot holds
    for (size_t b = 0; b < blocks.size(); b++) {
        block_word(b, b + 1 < blocks.size() ? b + 1 : npos);
        for (auto &cr : blocks[b].rows) cur_row[cr.first] = (int32_t)cr.second;
        for (size_t idx = blocks[b].first; idx < blocks[b].last; idx++) {
            const Instr &in = p.code[idx];
            struct Opnd {
                bool is_const;
                uint32_t v;
            };
            // (a forwarded slot operand is never read from LDS: its row field is left 0)
            auto opnd = [&](uint8_t kind, uint32_t payload, bool forwarded) -> Opnd {
                if (kind == K_SLOT) {
                    if (forwarded) return {false, 0};
                    const int32_t v = slot_value.at(payload);
                    if (v < 0 || value_row[(size_t)v] < 0) throw OvmFail(1, "stage_cached: internal (value not in a row)");
                    return {false, (uint32_t)value_row[(size_t)v]};
                }
                if (kind == K_COL) {
                    if (cur_row[payload] < 0) throw OvmFail(1, "stage_cached: internal (column not cached)");
                    return {false, (uint32_t)cur_row[payload]};
                }
                if (kind == K_CONST) return {true, payload};
                throw OvmFail(1, "stage_cached: bad operand kind");
            };
            auto emit = [&](uint32_t op, uint32_t dst, uint32_t a, uint32_t b2) { cp.words.push_back(StagedInstr{op, dst, a, b2}); };
            const bool f_a = (in.fwd & FWD_A) != 0, f_b = (in.fwd & FWD_B) != 0;
            const Opnd a = opnd(in.ak, in.a, f_a);
            if (in.op >= OP_ACC) {  // ACC, DEN (accumulator 0), LEAF
                // the value must be in the result register: forwarded from the instruction before, or moved there
                if (!f_a) {
                    if (a.is_const) emit(S_MOVC | S_B_CONST | S_NOSTORE, 0, 0, a.v);
                    else emit(S_MOV | S_B_CONST | S_NOSTORE, 0, a.v, 0);
                }
                if (in.op == OP_LEAF) {
                    while (cp.words.size() % 4 != 3) cp.words.push_back(nop);
                    emit(C_BLOCK | C_LEAF, in.b, 0, 0);
                    cp.n_leaves++;
                    continue;
                }
                if (!defer_weights && in.b >= weights.size()) throw OvmFail(1, "stage_cached: weight index out of range");
                const uint32_t acc = in.op == OP_ACC ? in.acc : 0;
                if (acc > 1) throw OvmFail(1, "stage_cached: accumulator out of range");
                const FpExt w = defer_weights ? FpExt::zero() : weights[in.b];
                cp.weight_at.push_back({(uint32_t)cp.words.size(), in.b});
                emit(C_ACC | w.c[0].v, (acc << 31) | w.c[1].v, w.c[2].v, w.c[3].v);
                continue;
            }
            const uint32_t f_ns = lv.stored[idx] ? 0 : S_NOSTORE;
            const uint32_t dst = lv.stored[idx] ? (uint32_t)value_row[idx] : 0;
            const Opnd bo = in.op == OP_NEG ? Opnd{true, 0} : opnd(in.bk, in.b, f_b);
            slot_value.at(in.dst) = (int32_t)idx;  // (after the operands were looked up: they name the values before)
            if (in.op == OP_NEG) {  // 0 - x
                if (a.is_const) emit(S_MOVC | S_B_CONST | f_ns, dst, 0, (-Fp::raw(a.v)).v);
                else emit(S_RSUB | S_B_CONST | (f_a ? S_FWD_A : 0) | f_ns, dst, a.v, 0);
                continue;
            }
            if (a.is_const && bo.is_const) {
                const Fp x = Fp::raw(a.v), y = Fp::raw(bo.v);
                const Fp r = in.op == OP_ADD ? x + y : in.op == OP_SUB ? x - y : x * y;
                emit(S_MOVC | S_B_CONST | f_ns, dst, 0, r.v);
            } else if (a.is_const) {  // constant on the left: commute, or b - a reversed (the program's b is the word's first operand)
                emit((in.op == OP_ADD ? S_ADD : in.op == OP_MUL ? S_MUL : S_RSUB) | S_B_CONST | (f_b ? S_FWD_A : 0) | f_ns, dst, bo.v, a.v);
            } else if (bo.is_const) {
                emit((in.op == OP_ADD ? S_ADD : in.op == OP_MUL ? S_MUL : S_SUB) | S_B_CONST | (f_a ? S_FWD_A : 0) | f_ns, dst, a.v, bo.v);
            } else if (f_b && !f_a) {  // the forwarded value is the SECOND operand: it becomes the first (a - result = S_RSUB)
                emit((in.op == OP_ADD ? S_ADD : in.op == OP_MUL ? S_MUL : S_RSUB) | S_FWD_A | f_ns, dst, 0, a.v);
            } else {
                emit((in.op == OP_ADD ? S_ADD : in.op == OP_MUL ? S_MUL : S_SUB) | (f_a ? S_FWD_A : 0) | ((f_a && f_b) ? S_FWD_B : 0) | f_ns,
                     dst, a.v, bo.v);
            }
        }
    }
    while (cp.words.size() % 8) cp.words.push_back(nop);  // (the kernel runs the groups of four two at a time)
    for (int k = 0; k < 4; k++) cp.words.push_back(nop);  // (... and fetches one group past the end)
    return cp;
}

// Host execution of a cached program on one row (self-test of stage_cached, no device; rows not scaled). Mirrors
// cached_vm.cuh word for word: the loads of a block word are requested (read from `row`) when the block word before it
// runs and are written when it runs itself; the first operand of an arithmetic word is LOADED INTO the result register.
template <class Leaf>
inline void cached_host_eval(const CachedProgram &cp, const std::vector<Fp> &row, FpExt out[2], Leaf &&leaf) {
    std::vector<Fp> lds(cp.num_rows, Fp::zero());
    Fp landing[CachedProgram::MAX_LOADS];
    Fp prev = Fp::zero();
    const Fp poison = Fp::from_canonical(0x5eed);
    const size_t n_words = cp.words.size() - 4;
    for (size_t i = 0; i < n_words; i++) {
        const StagedInstr &in = cp.words[i];
        if (in.op & C_ACC) {
            FpExt w;
            w.c[0] = Fp::raw(in.op & 0x7fffffffu), w.c[1] = Fp::raw(in.dst & 0x7fffffffu), w.c[2] = Fp::raw(in.a), w.c[3] = Fp::raw(in.b);
            out[in.dst >> 31] += w * prev;
        } else if ((in.op & C_BLOCK) && (in.op & C_LEAF)) {
            if (i % 4 != 3) throw OvmFail(1, "cached_host_eval: leaf marker not last in its group");
            leaf(in.dst, prev, out[0]);
            out[0] = FpExt::zero();
        } else if (in.op & C_BLOCK) {
            if (i % 4) throw OvmFail(1, "cached_host_eval: block word inside a group");
            const uint32_t n_land = in.op & 0xff, n_issue = (in.op >> 8) & 0xff;
            const uint32_t *rec = cp.loads.data() + (size_t)in.dst * 2 * CachedProgram::MAX_LOADS;
            for (uint32_t j = 0; j < n_land; j++) lds.at(rec[j]) = landing[j];
            for (uint32_t j = 0; j < n_issue; j++) landing[j] = row.at(rec[CachedProgram::MAX_LOADS + j]);
        } else {
            if ((in.op & S_FWD_B) && !(in.op & S_FWD_A)) throw OvmFail(1, "cached_host_eval: second operand forwarded alone");
            if (!(in.op & S_FWD_A) && !(in.op & S_MOVC)) prev = lds.at(in.a);
            const Fp a = prev;
            const Fp bv = (in.op & S_B_CONST) ? Fp::raw(in.b) : (in.op & S_FWD_B) ? prev : lds.at(in.b);
            Fp r;
            if (in.op & S_MUL) r = a * bv;
            else if (in.op & S_ADD) r = a + bv;
            else if (in.op & S_SUB) r = a - bv;
            else if (in.op & S_RSUB) r = bv - a;
            else if (in.op & S_MOV) r = a;
            else r = bv;
            if (!(in.op & S_NOSTORE)) lds.at(in.dst) = r;
            else if (in.dst) lds.at(in.dst) = poison;
            prev = r;
        }
    }
}
inline void cached_host_eval(const CachedProgram &cp, const std::vector<Fp> &row, FpExt out[2]) {
    cached_host_eval(cp, row, out, [](uint32_t, Fp, const FpExt &) { throw OvmFail(1, "cached_host_eval: unexpected LEAF"); });
}

// ---------------------------------------------------------------------------------------------
// Linear terms of an accumulation program, clustered by the columns they read.
//
// A round-0 program is a weighted sum  sum_i W[i] * root_i  and the roots of a real AIR are themselves sums: a 16-bit
// limb of Keccak-f's theta step is  sum_z 2^z * xor3(A'[z], C[z], C'[z]) - a, one root reading 49 columns, and each of
// those columns is read again by four or five other roots hundreds of constraints away (the real Keccak-f AIR: 2 859
// columns, 16 400 column reads per row). Whatever the executor stages per block is staged again for every root -- round 4
// measured 5.6x the coset LDE's bytes and showed that NO order of the ROOTS changes it.
// The sum is linear, so it is regrouped here: every root is pushed down through +, -, negation and multiplication by
// constants while the node it reaches still reads more than `support_cap` columns; what is left are TERMS (a product, a
// short expression, a column) with a base-field coefficient each, term t weighs  W'[t] = sum coef * W[i]  (one term can
// serve many roots), and the terms are ordered so that consecutive ones share columns: greedily, the next term is the
// one that brings the fewest columns not among the last `cache_rows` staged ones. The result is exact in the field --
// the same sum in another order -- and the Keccak-f program reads 4 600 columns per row instead of 16 000 with a cache
// of eight. Programs whose roots are all small come back unchanged (one term per root, in root order).
// ---------------------------------------------------------------------------------------------
struct TermPart {
    uint32_t term, weight;  // W'[term] += coef * W[weight]
    Fp coef;
};
struct TermRoots {
    std::vector<Root> roots;  // one OP_ACC root per term, clustered order; Root::weight = the term's index
    std::vector<TermPart> parts;
    bool split = false;  // some root was taken apart (else: the roots as they came)
    template <class E>
    std::vector<E> weights(const std::vector<E> &W) const {
        std::vector<E> out(std::max<size_t>(roots.size(), 1), E::zero());
        for (const TermPart &p : parts) {
            if (p.weight >= W.size()) throw OvmFail(1, "term weights: weight index out of range");
            out[p.term] = out[p.term] + W[p.weight] * p.coef;
        }
        return out;
    }
};
static constexpr uint32_t TERM_SUPPORT_CAP = 6, TERM_CLUSTER_ROWS = 8;
inline TermRoots linear_terms(const AirKey &air, const ColumnMap &cm, const std::vector<Fp> &public_values,
                              const std::vector<Root> &roots, uint32_t support_cap = TERM_SUPPORT_CAP,
                              uint32_t cache_rows = TERM_CLUSTER_ROWS) {
    TermRoots out;
    size_t n = 0;
    for (auto &r : roots) {
        if (r.op != OP_ACC || r.node == ROOT_ONE_NODE) throw OvmFail(1, "linear_terms: accumulation roots only");
        n = std::max(n, (size_t)r.node + 1);
    }
    if (n > air.nodes.size()) throw OvmFail(1, "linear_terms: root out of range");
    auto is_op = [&](uint32_t i) { return air.nodes[i].kind >= N_ADD; };
    auto binary = [&](uint32_t i) { return air.nodes[i].kind != N_NEG; };
    // reachable nodes
    std::vector<char> need(n, 0);
    {
        std::vector<uint32_t> stack;
        for (auto &r : roots)
            if (!need[r.node]) need[r.node] = 1, stack.push_back(r.node);
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (!is_op(i)) continue;
            const Node &nd = air.nodes[i];
            if (nd.left >= i || (binary(i) && nd.right >= i)) throw OvmFail(1, "linear_terms: DAG not topological");
            if (!need[nd.left]) need[nd.left] = 1, stack.push_back(nd.left);
            if (binary(i) && !need[nd.right]) need[nd.right] = 1, stack.push_back(nd.right);
        }
    }
    // per node: compile-time constant (value), or the columns it reads while they are at most support_cap (else `big`)
    auto column_of = [&](const Node &nd) -> int64_t {
        switch (nd.kind) {
            case N_VAR:
                if (nd.entry == E_MAIN) {
                    if (nd.part_index >= cm.part_width.size() || nd.index >= cm.part_width[nd.part_index] || (nd.offset && !cm.need_rot))
                        throw OvmFail(1, "main variable out of range");
                    return cm.col(nd.part_index, nd.offset, nd.index);
                }
                if (nd.entry == E_PREPROCESSED) {
                    if (nd.index >= cm.pre_width || (nd.offset && !cm.need_rot)) throw OvmFail(1, "preprocessed variable out of range");
                    return cm.pre_col(nd.offset, nd.index);
                }
                if (nd.entry == E_PUBLIC) return -1;
                throw OvmFail(801, "challenge variables not supported");
            case N_IS_FIRST: return 0;
            case N_IS_TRANSITION: return 1;
            case N_IS_LAST: return 2;
            default: return -1;
        }
    };
    std::vector<char> is_const(n, 0), big(n, 0);
    std::vector<Fp> cval(n, Fp::zero());
    std::vector<std::vector<uint32_t>> sup(n);
    for (size_t i = 0; i < n; i++) {
        if (!need[i]) continue;
        const Node &nd = air.nodes[i];
        if (!is_op((uint32_t)i)) {
            const int64_t c = column_of(nd);
            if (c >= 0) {
                sup[i].push_back((uint32_t)c);
            } else {
                is_const[i] = 1;
                cval[i] = nd.kind == N_CONST ? nd.constant : public_values.at(nd.index);
            }
            continue;
        }
        const bool bin = binary((uint32_t)i);
        if (is_const[nd.left] && (!bin || is_const[nd.right])) {
            is_const[i] = 1;
            const Fp x = cval[nd.left], y = bin ? cval[nd.right] : Fp::zero();
            cval[i] = nd.kind == N_ADD ? x + y : nd.kind == N_SUB ? x - y : nd.kind == N_MUL ? x * y : -x;
            continue;
        }
        big[i] = big[nd.left] || (bin && big[nd.right]);
        if (!big[i]) {
            if (bin) std::set_union(sup[nd.left].begin(), sup[nd.left].end(), sup[nd.right].begin(), sup[nd.right].end(), std::back_inserter(sup[i]));
            else sup[i] = sup[nd.left];
            if (sup[i].size() > support_cap) big[i] = 1, sup[i].clear();
        }
    }
    // a big node that is a sum, a negation or a product with a constant is taken apart
    auto splittable = [&](uint32_t i) {
        if (!big[i]) return false;
        const Node &nd = air.nodes[i];
        return nd.kind == N_ADD || nd.kind == N_SUB || nd.kind == N_NEG || (nd.kind == N_MUL && (is_const[nd.left] || is_const[nd.right]));
    };
    bool any_split = false;
    for (auto &r : roots) any_split |= splittable(r.node);
    if (!any_split) {
        for (size_t k = 0; k < roots.size(); k++) {
            out.roots.push_back(Root{roots[k].node, OP_ACC, roots[k].acc, (uint32_t)k});
            out.parts.push_back(TermPart{(uint32_t)k, roots[k].weight, Fp::one()});
        }
        return out;
    }
    out.split = true;
    // terms in first-appearance order: key (node, accumulator); the constant-one term per accumulator
    struct Term {
        uint32_t node;
        uint8_t acc;
        std::vector<uint32_t> cols;
    };
    std::vector<Term> terms;
    std::vector<int32_t> term_of[2] = {std::vector<int32_t>(n, -1), std::vector<int32_t>(n, -1)};
    int32_t one_term[2] = {-1, -1};
    std::vector<TermPart> parts;
    std::vector<Fp> coef(n, Fp::zero());
    std::vector<char> queued(n, 0);
    std::vector<uint32_t> heap;  // max-heap of node indices: a node is handled after everything that can feed it
    auto feed = [&](uint32_t i, Fp c) {
        coef[i] = coef[i] + c;
        if (!queued[i]) {
            queued[i] = 1;
            heap.push_back(i);
            std::push_heap(heap.begin(), heap.end());
        }
    };
    for (auto &r : roots) {
        if (r.acc > 1) throw OvmFail(1, "linear_terms: accumulator out of range");
        feed(r.node, Fp::one());
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end());
            const uint32_t i = heap.back();
            heap.pop_back();
            const Fp c = coef[i];
            coef[i] = Fp::zero();
            queued[i] = 0;
            if (c.v == 0) continue;
            const Node &nd = air.nodes[i];
            if (is_const[i]) {
                if (cval[i].v == 0) continue;
                if (one_term[r.acc] < 0) {
                    one_term[r.acc] = (int32_t)terms.size();
                    terms.push_back(Term{ROOT_ONE_NODE, r.acc, {}});
                }
                parts.push_back(TermPart{(uint32_t)one_term[r.acc], r.weight, c * cval[i]});
            } else if (splittable(i)) {
                if (nd.kind == N_MUL) {
                    const bool lc = is_const[nd.left];
                    feed(lc ? nd.right : nd.left, c * cval[lc ? nd.left : nd.right]);
                } else {
                    feed(nd.left, nd.kind == N_NEG ? -c : c);
                    if (nd.kind != N_NEG) feed(nd.right, nd.kind == N_SUB ? -c : c);
                }
            } else {
                int32_t &t = term_of[r.acc][i];
                if (t < 0) {
                    t = (int32_t)terms.size();
                    terms.push_back(Term{i, r.acc, {}});
                }
                parts.push_back(TermPart{(uint32_t)t, r.weight, c});
            }
        }
    }
    // the columns a term reads (a big term: collected from its expression)
    {
        std::vector<uint32_t> seen(n, 0), stack;
        for (size_t t = 0; t < terms.size(); t++) {
            Term &tm = terms[t];
            if (tm.node == ROOT_ONE_NODE) continue;
            if (!big[tm.node]) {
                tm.cols = sup[tm.node];
                continue;
            }
            stack.assign(1, tm.node);
            while (!stack.empty()) {
                const uint32_t i = stack.back();
                stack.pop_back();
                if (seen[i] == t + 1) continue;
                seen[i] = (uint32_t)t + 1;
                const Node &nd = air.nodes[i];
                if (!is_op(i)) {
                    const int64_t c = column_of(nd);
                    if (c >= 0) tm.cols.push_back((uint32_t)c);
                } else {
                    stack.push_back(nd.left);
                    if (binary(i)) stack.push_back(nd.right);
                }
            }
            std::sort(tm.cols.begin(), tm.cols.end());
            tm.cols.erase(std::unique(tm.cols.begin(), tm.cols.end()), tm.cols.end());
        }
    }
    // greedy order: the next term is the undone one, among those that read a cached column, with the fewest columns to
    // bring (ties: the one whose cached columns are closest to their last use); nothing cached is read any more -> the
    // next undone term in first-appearance order. The cache model is least-recently-used over `cache_rows` columns.
    const size_t T = terms.size();
    std::vector<uint32_t> order;
    order.reserve(T);
    {
        std::vector<uint32_t> remaining(cm.total, 0);
        std::vector<std::vector<uint32_t>> col_terms(cm.total);
        for (size_t t = 0; t < T; t++)
            for (uint32_t c : terms[t].cols) remaining[c]++, col_terms[c].push_back((uint32_t)t);
        std::vector<char> done(T, 0), cached(cm.total, 0);
        std::vector<uint32_t> cache;  // least recently used first
        std::vector<uint32_t> stamp(T, 0);
        size_t cursor = 0;
        for (size_t step = 1; step <= T; step++) {
            int64_t best = -1;
            uint32_t best_new = ~0u;
            double best_score = -1;
            for (uint32_t c : cache)
                for (uint32_t t : col_terms[c]) {
                    if (done[t] || stamp[t] == step) continue;
                    stamp[t] = (uint32_t)step;
                    uint32_t fresh = 0;
                    double score = 0;
                    for (uint32_t x : terms[t].cols) {
                        if (cached[x]) score += 1.0 / remaining[x];
                        else fresh++;
                    }
                    if (fresh < best_new || (fresh == best_new && score > best_score)) best = t, best_new = fresh, best_score = score;
                }
            if (best < 0) {
                while (done[cursor]) cursor++;
                best = (int64_t)cursor;
            }
            done[best] = 1;
            order.push_back((uint32_t)best);
            for (uint32_t x : terms[best].cols) {
                remaining[x]--;
                if (cached[x]) cache.erase(std::find(cache.begin(), cache.end(), x));
                cache.push_back(x);
                cached[x] = 1;
            }
            // columns nothing will read again leave first, then the least recently used
            for (size_t k = 0; k < cache.size();) {
                if (remaining[cache[k]] == 0) cached[cache[k]] = 0, cache.erase(cache.begin() + k);
                else k++;
            }
            while (cache.size() > cache_rows) cached[cache.front()] = 0, cache.erase(cache.begin());
        }
    }
    std::vector<uint32_t> new_index(T);
    for (size_t k = 0; k < T; k++) {
        new_index[order[k]] = (uint32_t)k;
        out.roots.push_back(Root{terms[order[k]].node, OP_ACC, terms[order[k]].acc, (uint32_t)k});
    }
    for (TermPart &p : parts) p.term = new_index[p.term];
    out.parts = std::move(parts);
    return out;
}

// The same roots split into `n_groups` independent sub-programs (consecutive runs of roots). All
// root ops here are additive (ACC), so the groups' outputs simply add; shared DAG nodes are
// recomputed per group. Used by the multilinear rounds, where the number of evaluation points
// shrinks every round and one lane interpreting the whole DAG serially is latency-bound: groups
// (and the X points) become extra grid dimensions instead.
struct GroupedProgram {
    std::vector<Instr> code;
    std::vector<uint32_t> off;  // n_groups + 1 offsets into code
    uint32_t max_slots = 1;  // slots an executor that ignores the forwarding hints needs (the stored ones + one scratch slot)
    uint32_t lds_slots = 1;  // slots of the values that are really stored: what an executor that honours the hints needs
};

// Slots renumbered by what is STORED. compile_program gives every result a slot, also the ones that live in the result
// register only (FWD_NOSTORE: read by the next instruction through the register). An executor that honours the hints
// (the LDS-slot kernel of the multilinear rounds: D x 4 KiB of LDS per slot and workgroup) needs rows only for values that
// are written. Here the slots are re-coloured by live range: a stored value (from the instruction that computes it to its
// last read that is not forwarded) takes the lowest free slot; a register-only value -- which an executor that ignores
// the hints still writes and reads back one instruction later -- then goes into a slot no stored value occupies during
// those two instructions, else into ONE scratch slot on top. Returns the number of slots of the stored values (what the
// hint-honouring executor needs); p.num_slots becomes what an executor that ignores the hints needs (never more than
// before). The real Keccak-f groups: 5 -> 4 stored.
inline uint32_t compact_slots(Program &p) {
    const size_t n = p.code.size();
    auto is_binary = [](const Instr &in) { return in.op == OP_ADD || in.op == OP_SUB || in.op == OP_MUL; };
    std::vector<int32_t> def(p.num_slots + 1, -1), last_read(n, -1);
    for (size_t i = 0; i < n; i++) {
        const Instr &in = p.code[i];
        if (in.ak == K_SLOT && !(in.fwd & FWD_A) && def.at(in.a) >= 0) last_read[def[in.a]] = (int32_t)i;
        if (is_binary(in) && in.bk == K_SLOT && !(in.fwd & FWD_B) && def.at(in.b) >= 0) last_read[def[in.b]] = (int32_t)i;
        if (in.op <= OP_NEG) def.at(in.dst) = (int32_t)i;
    }
    // pass 1: the stored values, lowest free slot first
    std::vector<uint32_t> new_of_value(n, ~0u);  // the new slot of the value instruction i computes
    std::vector<size_t> dies(n, 0);
    uint32_t stored = 0;
    {
        std::vector<std::vector<uint32_t>> dying(n + 1);
        std::vector<uint32_t> free_ids;
        for (size_t i = 0; i < n; i++) {
            const Instr &in = p.code[i];
            if (in.op <= OP_NEG && !(in.fwd & FWD_NOSTORE)) {
                uint32_t id;
                if (free_ids.empty()) {
                    id = stored++;
                } else {
                    auto it = std::min_element(free_ids.begin(), free_ids.end());
                    id = *it;
                    free_ids.erase(it);
                }
                new_of_value[i] = id;
                dies[i] = std::max<size_t>(i, last_read[i] < 0 ? 0 : (size_t)last_read[i]);
                dying[dies[i]].push_back(id);
            }
            for (uint32_t id : dying[i]) free_ids.push_back(id);
        }
    }
    // pass 2: a register-only value (written at i, read back at i + 1 by an executor that ignores the hints) goes where no
    // stored value lives during those two instructions -- the highest such slot --, else into one scratch slot on top
    std::vector<std::vector<std::pair<size_t, size_t>>> busy(stored);
    for (size_t i = 0; i < n; i++)
        if (new_of_value[i] != ~0u) busy[new_of_value[i]].push_back({i, dies[i]});
    bool scratch = false;
    for (size_t i = 0; i < n; i++) {
        const Instr &in = p.code[i];
        if (!(in.op <= OP_NEG && (in.fwd & FWD_NOSTORE))) continue;
        uint32_t id = stored;
        for (uint32_t k = stored; k-- > 0;) {
            bool free_here = true;
            for (auto &iv : busy[k])
                if (iv.first <= i + 1 && iv.second >= i) {
                    free_here = false;
                    break;
                }
            if (free_here) {
                id = k;
                break;
            }
        }
        scratch |= id == stored;
        new_of_value[i] = id;
    }
    // rename
    std::vector<int32_t> value_in(p.num_slots + 1, -1);  // the instruction whose result an OLD slot holds
    for (size_t i = 0; i < n; i++) {
        Instr &in = p.code[i];
        auto renamed = [&](uint32_t old) -> uint32_t {
            const int32_t v = value_in.at(old);
            return v < 0 ? 0 : new_of_value[(size_t)v];
        };
        if (in.ak == K_SLOT) in.a = renamed(in.a);
        if (is_binary(in) && in.bk == K_SLOT) in.b = renamed(in.b);
        if (in.op <= OP_NEG) {
            value_in.at(in.dst) = (int32_t)i;
            in.dst = (uint16_t)new_of_value[i];
        }
    }
    p.num_slots = std::max<uint32_t>(stored + (scratch ? 1 : 0), 1);
    return std::max<uint32_t>(stored, 1);
}

// `slot_target`: the groups are what the multilinear rounds interpret on extension-field values, D points at a time,
// with their slots in LDS (stages.hip: mle_round_points_lds_kernel): a group qualifies with at most 19 / D slots, so the
// compiler is asked for that few (it recomputes cheap shared nodes until the group fits or its code would double).
inline uint32_t mle_slot_target(uint32_t points) { return std::max<uint32_t>(2, 19 / std::max<uint32_t>(points, 1)); }
inline GroupedProgram compile_grouped(const AirKey &air, const ColumnMap &cm, const std::vector<Fp> &public_values,
                                      const std::vector<Root> &roots, uint32_t num_weights, uint32_t n_groups,
                                      uint32_t slot_target = PROGRAM_SLOT_TARGET) {
    GroupedProgram g;
    if (n_groups < 1) n_groups = 1;
    if (n_groups > roots.size()) n_groups = roots.empty() ? 1 : (uint32_t)roots.size();
    // Groups are runs of consecutive roots of about equal COST (instructions of the roots' expression trees), not of equal
    // root count: the real Keccak-f AIR mixes 3-instruction boolean checks with 330-instruction limb constraints, and with
    // 25 roots per group the longest group was ~100x the shortest -- the grouped kernels ran at 0.6 waves per SIMD waiting
    // for it (SQ counters, round 3).
    std::vector<uint64_t> prefix(roots.size() + 1, 0);
    {
        size_t n = 0;
        for (auto &r : roots) n = std::max(n, (size_t)r.node + 1);
        if (n > air.nodes.size()) throw OvmFail(1, "program compiler: root out of range");
        // cost of a root = the distinct operation nodes its expression reaches (what a group of its own would emit,
        // give or take recomputation); a tree count would double at every `x + x`
        std::vector<uint32_t> seen(n, 0);
        std::vector<uint32_t> stack;
        for (size_t r = 0; r < roots.size(); r++) {
            const uint32_t stamp = (uint32_t)r + 1;
            uint64_t c = 1;
            stack.assign(1, roots[r].node);
            while (!stack.empty()) {
                const uint32_t i = stack.back();
                stack.pop_back();
                const Node &nd = air.nodes[i];
                if (nd.kind < N_ADD || seen[i] == stamp) continue;
                seen[i] = stamp;
                c++;
                stack.push_back(nd.left);
                if (nd.kind != N_NEG) stack.push_back(nd.right);
            }
            prefix[r + 1] = prefix[r] + c;
        }
    }
    g.off.push_back(0);
    size_t lo = 0;
    for (uint32_t k = 0; k < n_groups; k++) {
        // the group ends where the cost prefix passes its share; every group gets at least one root
        size_t hi = lo + 1;
        const uint64_t want = prefix.back() * (k + 1) / n_groups;
        while (hi < roots.size() && prefix[hi] < want && roots.size() - hi > n_groups - 1 - k) hi++;
        if (k + 1 == n_groups) hi = roots.size();
        if (lo >= roots.size()) {  // (fewer roots than groups cannot happen: n_groups <= roots.size())
            g.off.push_back((uint32_t)g.code.size());
            continue;
        }
        std::vector<Root> sub(roots.begin() + lo, roots.begin() + hi);
        Program p = compile_program(air, cm, public_values, sub, num_weights, slot_target);
        g.lds_slots = std::max(g.lds_slots, compact_slots(p));
        g.code.insert(g.code.end(), p.code.begin(), p.code.end());
        g.off.push_back((uint32_t)g.code.size());
        g.max_slots = std::max(g.max_slots, p.num_slots);
        lo = hi;
    }
    return g;
}

// ---------------------------------------------------------------------------------------------
// Interpreter (host and device). `Fetch(col) -> T` supplies trace cells at the current point.
// ---------------------------------------------------------------------------------------------
template <class T>
struct Lift;
template <>
struct Lift<Fp> {
    BB_HD static Fp of(uint32_t mont) { return Fp::raw(mont); }
};
template <>
struct Lift<FpExt> {
    BB_HD static FpExt of(uint32_t mont) { return FpExt::from_base(Fp::raw(mont)); }
};

template <class T, class Fetch>
BB_HD T load_operand(uint8_t kind, uint32_t payload, const T *slots, Fetch &fetch) {
    if (kind == K_SLOT) return slots[payload];
    if (kind == K_COL) return fetch(payload);
    return Lift<T>::of(payload);
}

// Runs `code`; accumulates into out[0..2) (ACC mode). `slots` is caller-provided scratch.
template <class T, class Fetch>
BB_HD void run_program_acc(const Instr *code, uint32_t n_instr, const FpExt *weights, T *slots, Fetch &fetch,
                           FpExt *out) {
    for (uint32_t pc = 0; pc < n_instr; pc++) {
        const Instr in = code[pc];
        T a = load_operand<T>(in.ak, in.a, slots, fetch);
        switch (in.op) {
            case OP_ADD: slots[in.dst] = a + load_operand<T>(in.bk, in.b, slots, fetch); break;
            case OP_SUB: slots[in.dst] = a - load_operand<T>(in.bk, in.b, slots, fetch); break;
            case OP_MUL: slots[in.dst] = a * load_operand<T>(in.bk, in.b, slots, fetch); break;
            case OP_NEG: slots[in.dst] = -a; break;
            default: out[in.acc] += weights[in.b] * a; break;  // OP_ACC
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Accumulator form of a grouped program (the LDS-slot kernels of the multilinear rounds, mle_vm.cuh: MleVm<D>::run).
// One RESULT REGISTER set holds the value of the word before; every word has at most one memory operand:
//     LOAD  acc = operand                     ADD / SUB / MUL  acc = acc op operand        RSUB  acc = operand - acc
//     ACC   out[k] += W[z] * (operand | acc)  NOP
// operand: a slot (an LDS row), a column (the pair of the lane's row, all points made from it), a base-field constant
// (only limb 0 is touched: no broadcast), or the result register itself. An instruction whose first operand is the
// result before it (FWD_A) is one word; with FWD_B the operands change places (SUB becomes RSUB); otherwise a LOAD word
// goes in front. `store`: the result is also written to its slot (absent on FWD_NOSTORE results). What the compiler
// made of the same machine written over Instr -- a copy of the result registers into `prev` and back per instruction,
// byte-wide instruction fetches through the vector memory path with a wait for ALL loads in flight in front of every
// decode, a 4 x D-word broadcast per constant -- was 178 VALU instructions per Keccak-f instruction where the arithmetic
// needs about 100.
// A group's words: a multiple of MLE_WORD_GROUP x 2 (NOP-padded; the executor alternates two landing register sets over
// groups of MLE_WORD_GROUP words), followed by 2 groups of NOPs (the word fetch runs two groups ahead).
// ---------------------------------------------------------------------------------------------
struct MleWord {
    uint32_t x;  // [2:0] op, [4:3] operand kind, [5] store, [6] accumulator index, [31:16] destination slot
    uint32_t y;  // operand payload: slot / flat column / Montgomery constant
    uint32_t z;  // ACC: weight index
    uint32_t w;
};
enum : uint32_t { W_LOAD = 0, W_ADD, W_SUB, W_RSUB, W_MUL, W_ACC, W_NOP = 7 };
enum : uint32_t { WK_SLOT = 0, WK_COL, WK_CONST, WK_SELF };
static constexpr uint32_t MLE_WORD_GROUP = 4;
struct MleWords {
    std::vector<MleWord> words;
    std::vector<uint32_t> off;  // n_groups + 1 offsets into words (a group's executable words: off[g+1] - off[g] - 2 groups)
    uint32_t rows = 1;          // slots the executor needs
};
// a weight in the form the executor's multiply-adds take it: b0..b3, 11 b1..11 b3 (bb31.h FpExt::Pre); 32 bytes = one
// s_load_dwordx8
struct MleWeight {
    uint32_t b[4], w[3], pad;
};
inline MleWeight mle_weight_of(const FpExt &e) {
    const FpExt::Pre p = e.pre();
    return MleWeight{{p.b[0].v, p.b[1].v, p.b[2].v, p.b[3].v}, {p.w[0].v, p.w[1].v, p.w[2].v}, 0};
}
inline MleWords encode_mle_words(const GroupedProgram &gp, bool honour_hints = true) {
    MleWords out;
    out.rows = honour_hints ? gp.lds_slots : gp.max_slots;
    out.off.push_back(0);
    auto word = [](uint32_t op, uint32_t kind, uint32_t payload, bool store = false, uint32_t dst = 0, uint32_t k = 0,
                   uint32_t widx = 0) {
        return MleWord{op | kind << 3 | (store ? 32u : 0u) | (k & 1) << 6 | dst << 16, payload, widx, 0};
    };
    auto kind_of = [](uint8_t k) -> uint32_t { return k == K_SLOT ? WK_SLOT : k == K_COL ? WK_COL : WK_CONST; };
    for (size_t g = 0; g + 1 < gp.off.size(); g++) {
        for (uint32_t i = gp.off[g]; i < gp.off[g + 1]; i++) {
            const Instr &in = gp.code[i];
            const uint16_t fwd = honour_hints ? in.fwd : 0;
            const bool fa = fwd & FWD_A, fb = in.op <= OP_MUL && (fwd & FWD_B);
            const bool store = !(fwd & FWD_NOSTORE);
            if (in.op == OP_ACC) {
                out.words.push_back(fa ? word(W_ACC, WK_SELF, 0, false, 0, in.acc, in.b)
                                       : word(W_ACC, kind_of(in.ak), in.a, false, 0, in.acc, in.b));
            } else if (in.op == OP_NEG) {  // -a = 0 - a
                if (!fa) out.words.push_back(word(W_LOAD, kind_of(in.ak), in.a));
                out.words.push_back(word(W_RSUB, WK_CONST, 0, store, in.dst));
            } else if (in.op <= OP_MUL) {
                const uint32_t op = in.op == OP_ADD ? W_ADD : in.op == OP_SUB ? W_SUB : W_MUL;
                const uint32_t rop = in.op == OP_SUB ? W_RSUB : op;  // the operands changed places
                if (fa && fb) {
                    out.words.push_back(word(op, WK_SELF, 0, store, in.dst));
                } else if (fa) {
                    out.words.push_back(word(op, kind_of(in.bk), in.b, store, in.dst));
                } else if (fb) {
                    out.words.push_back(word(rop, kind_of(in.ak), in.a, store, in.dst));
                } else if (in.ak == K_CONST && in.bk != K_CONST) {  // the constant as the word's operand: limb 0 only
                    out.words.push_back(word(W_LOAD, kind_of(in.bk), in.b));
                    out.words.push_back(word(rop, WK_CONST, in.a, store, in.dst));
                } else {
                    out.words.push_back(word(W_LOAD, kind_of(in.ak), in.a));
                    out.words.push_back(word(op, kind_of(in.bk), in.b, store, in.dst));
                }
            } else {
                throw OvmFail(1, "encode_mle_words: not an accumulation program");
            }
        }
        // (an empty group still runs one pair of word groups: the executor counts down from at least two)
        while (out.words.size() == out.off.back() || (out.words.size() - out.off.back()) % (2 * MLE_WORD_GROUP))
            out.words.push_back(word(W_NOP, WK_CONST, 0));
        for (uint32_t k = 0; k < 2 * MLE_WORD_GROUP; k++) out.words.push_back(word(W_NOP, WK_CONST, 0));
        out.off.push_back((uint32_t)out.words.size());
    }
    return out;
}
// Host evaluation of one group's words at one point (the self-test's check of the encoding; the device executor makes
// D points of every column operand and is otherwise this loop).
template <class T, class Fetch>
inline void mle_words_host_eval(const MleWord *w, uint32_t n_words, const FpExt *weights, T *slots, Fetch &fetch, FpExt *out) {
    T acc = Lift<T>::of(0);
    for (uint32_t i = 0; i < n_words; i++) {
        const uint32_t op = w[i].x & 7, kind = (w[i].x >> 3) & 3;
        if (op == W_NOP) continue;
        const T v = kind == WK_SLOT ? slots[w[i].y] : kind == WK_COL ? fetch(w[i].y) : kind == WK_CONST ? Lift<T>::of(w[i].y) : acc;
        switch (op) {
            case W_LOAD: acc = v; break;
            case W_ADD: acc = acc + v; break;
            case W_SUB: acc = acc - v; break;
            case W_RSUB: acc = v - acc; break;
            case W_MUL: acc = acc * v; break;
            default: out[(w[i].x >> 6) & 1] += weights[w[i].z] * v; break;  // W_ACC
        }
        if (w[i].x & 32u) slots[w[i].x >> 16] = acc;
    }
}

// The same for R rows per lane at once (base-field programs): one instruction decode -- the
// interpreter's dominant cost, ~30 scalar instructions -- is amortised over R field operations.
// slots[s][r], out[r][acc]; fetch(col, r) reads column `col` at the lane's r-th row.
template <int R, class Fetch>
BB_HD void load_operand_rows(uint8_t kind, uint32_t payload, const Fp (*slots)[R], Fetch &fetch, Fp (&v)[R]) {
    if (kind == K_SLOT) {
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = slots[payload][r];
    } else if (kind == K_COL) {
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = fetch(payload, r);
    } else {
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = Fp::raw(payload);
    }
}
template <int R, class Fetch>
BB_HD void run_program_acc_rows(const Instr *code, uint32_t n_instr, const FpExt *weights, Fp (*slots)[R], Fetch &fetch,
                                FpExt (*out)[2]) {
    for (uint32_t pc = 0; pc < n_instr; pc++) {
        const Instr in = code[pc];
        Fp a[R], b[R];
        load_operand_rows<R>(in.ak, in.a, slots, fetch, a);
        switch (in.op) {
            case OP_ADD:
                load_operand_rows<R>(in.bk, in.b, slots, fetch, b);
#pragma unroll
                for (int r = 0; r < R; r++) slots[in.dst][r] = a[r] + b[r];
                break;
            case OP_SUB:
                load_operand_rows<R>(in.bk, in.b, slots, fetch, b);
#pragma unroll
                for (int r = 0; r < R; r++) slots[in.dst][r] = a[r] - b[r];
                break;
            case OP_MUL:
                load_operand_rows<R>(in.bk, in.b, slots, fetch, b);
#pragma unroll
                for (int r = 0; r < R; r++) slots[in.dst][r] = a[r] * b[r];
                break;
            case OP_NEG:
#pragma unroll
                for (int r = 0; r < R; r++) slots[in.dst][r] = -a[r];
                break;
            default: {  // OP_ACC
                const FpExt w = weights[in.b];
                if (in.acc) {  // (static indices: a runtime index would push `out` into scratch memory)
#pragma unroll
                    for (int r = 0; r < R; r++) out[r][1] += w * a[r];
                } else {
#pragma unroll
                    for (int r = 0; r < R; r++) out[r][0] += w * a[r];
                }
                break;
            }
        }
    }
}

// Leaf mode: `emit(k, numer, denom)` is called per interaction.
template <class T, class Fetch, class Emit>
BB_HD void run_program_leaves(const Instr *code, uint32_t n_instr, const FpExt *weights, const FpExt *leaf_const,
                              T *slots, Fetch &fetch, Emit &emit) {
    FpExt den = FpExt::zero();
    for (uint32_t pc = 0; pc < n_instr; pc++) {
        const Instr in = code[pc];
        T a = load_operand<T>(in.ak, in.a, slots, fetch);
        switch (in.op) {
            case OP_ADD: slots[in.dst] = a + load_operand<T>(in.bk, in.b, slots, fetch); break;
            case OP_SUB: slots[in.dst] = a - load_operand<T>(in.bk, in.b, slots, fetch); break;
            case OP_MUL: slots[in.dst] = a * load_operand<T>(in.bk, in.b, slots, fetch); break;
            case OP_NEG: slots[in.dst] = -a; break;
            case OP_DEN: den += weights[in.b] * a; break;
            default:  // OP_LEAF
                emit(in.b, a, den + leaf_const[in.b]);
                den = FpExt::zero();
                break;
        }
    }
}

}  // namespace ovm
