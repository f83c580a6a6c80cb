// LogUp-GKR fractional sumcheck over a fraction tree whose LEAVES ARE SPLIT INTO CONTIGUOUS RANGES,
// one per rank, plus an analytic range of padding leaves. Host side: the driver that every rank
// runs in lock step, and the bookkeeping of the few tree nodes / table records that straddle two
// ranges. Protocol: crates/stark-backend/src/prover/logup_zerocheck/fractional_sumcheck_gkr.rs:60-213
// (one global heap-ordered tree, children 2i, 2i+1; layer j proved by a sumcheck over j-1
// variables, low index bit bound first); which part of it runs where is this backend's design
// ("each GPU builds the sub-tree of its aligned leaf range; the top layers and
// every layer's sumcheck round sums are combined on the host").
//
// Ownership. Leaves [0, 2^TR) are tiled by owner ranges in rank order (a rank's traces are
// consecutive in sorted order, so its interaction slices are one run), then the padding run
// [used, 2^TR) of default leaves (0, alpha). Rank o owns fraction i of layer j iff the whole leaf
// span of i lies in its range: F[j-1][o] = [ceil(lo/2), floor(hi/2)) of F[j][o]. Fractions owned by
// nobody ("boundary", at most one per range end and layer) are computed by EVERY rank on the host
// from the ranges' edge fractions, which are exchanged once after the build.
//
// A layer's sumcheck works on a table of records (record y = fractions 2y, 2y+1 of the layer);
// sub-round sr pairs records (2y, 2y+1), low bit first -- the same halving structure as the tree.
// A rank sums its wholly-owned pairs on the device; a pair that straddles ranges is summed by every
// rank on the host from the two unpaired edge records, which travel with the partial sums of that
// sub-round (one small all-gather per sub-round carries both). Folding keeps the structure: the
// folded host pairs are the next sub-round's host records.
//
// Padding never touches a device: a padding-only subtree of height h is (0, alpha^(2^h)), a table
// record of such nodes stays constant under folding, and the sum of the eq weights over a run of
// indices is a prefix product -- so its contribution to every round is a closed form. (For the
// 1-GPU config D proof that removes the 6/32 of the GKR work that was padding.)
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <functional>
#include <map>

#include "host.h"
#include "shard.h"

namespace ovm {

struct Frac {
    EF p, q;
};
inline Frac frac_add(const Frac &a, const Frac &b) { return Frac{a.p * b.q + a.q * b.p, a.q * b.q}; }
struct GkrRec {
    EF v[4];  // p0, q0, p1, q1 = fractions 2y, 2y+1
};
inline GkrRec rec_of(const Frac &a, const Frac &b) { return GkrRec{{a.p, a.q, b.p, b.q}}; }
inline GkrRec rec_fold(const GkrRec &a, const GkrRec &b, const EF &r) {
    GkrRec o;
    for (int k = 0; k < 4; k++) o.v[k] = a.v[k] + (b.v[k] - a.v[k]) * r;
    return o;
}
// sums[X-1] += e * (p0 q1 + p1 q0 + lambda q0 q1)(X), every quantity the line through its values on a
// (X = 0) and b (X = 1): what the device kernels (stages.hip: gkr_pair_sums) add per pair
inline void rec_pair_sums(const GkrRec &a, const GkrRec &b, const EF &e, const EF &lambda, unsigned n_points, EF *sums) {
    EF v[4], d[4];
    for (int k = 0; k < 4; k++) {
        v[k] = b.v[k];  // X = 1
        d[k] = b.v[k] - a.v[k];
    }
    for (unsigned x = 1; x <= n_points; x++) {
        sums[x - 1] += (v[0] * v[3] + v[1] * (v[2] + lambda * v[3])) * e;
        for (int k = 0; k < 4; k++) v[k] += d[k];
    }
}

// The same over a batch of pairs, vectorised on hosts with AVX2 (host_gkr_batch.cpp): sums[x-1] += sum_i e[i] * (...)(x);
// out[i] = rec_fold(A[i], B[i], r); out[i] = eq_at(pt, nv, y[i]). Bit-identical to the scalar functions.
void gkr_pair_sums_batch(const GkrRec *A, const GkrRec *B, const EF *e, size_t n, const EF &lambda, unsigned n_points, EF *sums);
void gkr_fold_batch(const GkrRec *A, const GkrRec *B, size_t n, const EF &r, GkrRec *out);

struct URange {
    uint64_t lo = 0, hi = 0;
    bool empty() const { return lo >= hi; }
    uint64_t size() const { return empty() ? 0 : hi - lo; }
    bool has(uint64_t i) const { return i >= lo && i < hi; }
};
inline URange up(URange r) {
    URange o{(r.lo + 1) >> 1, r.hi >> 1};
    if (r.empty() || o.lo >= o.hi) o = URange{0, 0};
    return o;
}
// the unpaired records of a range (a left end at an odd index, a right end at an even one)
inline unsigned range_edges(URange r, uint64_t out[2]) {
    unsigned n = 0;
    if (r.empty()) return 0;
    if (r.lo & 1) out[n++] = r.lo;
    if ((r.hi & 1) && !(n && out[0] == r.hi - 1)) out[n++] = r.hi - 1;
    return n;
}

// sum_{y < n} eq(point[0..nv), y), y's bit k paired with point[k]
inline EF eq_prefix_sum(const EF *point, unsigned nv, uint64_t n) {
    if (n >> nv) return EF::one();
    EF acc = EF::zero(), pre = EF::one();
    for (unsigned k = nv; k-- > 0;) {
        if ((n >> k) & 1) {
            acc += pre * (EF::one() - point[k]);
            pre = pre * point[k];
        } else {
            pre = pre * (EF::one() - point[k]);
        }
    }
    return acc;
}
inline EF eq_at(const EF *point, unsigned nv, uint64_t y) {
    EF acc = EF::one();
    for (unsigned k = 0; k < nv; k++) acc = acc * eq1b(point[k], (y >> k) & 1);
    return acc;
}

void gkr_eq_batch(const EF *point, unsigned nv, const uint64_t *y, size_t n, EF *out);

// Who owns what. Owners 0..world-1 are the ranks, owner `world` is the padding run (maybe empty).
struct GkrPlan {
    unsigned TR = 0;
    uint32_t world = 1;
    std::vector<std::vector<URange>> F;  // F[j][o], j = 0..TR
    GkrPlan() {}
    GkrPlan(unsigned TR_, const std::vector<URange> &rank_leaves, uint64_t used) : TR(TR_), world((uint32_t)rank_leaves.size()) {
        const uint64_t n_leaves = uint64_t(1) << TR;
        std::vector<URange> leaf = rank_leaves;
        leaf.push_back(used < n_leaves ? URange{used, n_leaves} : URange{0, 0});
        uint64_t at = 0;
        for (auto &r : leaf) {
            if (r.empty()) continue;
            if (r.lo != at) throw OvmFail(1, "internal: GKR leaf ranges do not tile the tree");
            at = r.hi;
        }
        if (at != n_leaves) throw OvmFail(1, "internal: GKR leaf ranges do not cover the tree");
        F.assign(TR + 1, std::vector<URange>(world + 1));
        F[TR] = leaf;
        for (unsigned j = TR; j > 0; j--)
            for (uint32_t o = 0; o <= world; o++) F[j - 1][o] = up(F[j][o]);
    }
    uint32_t owners() const { return world + 1; }
    int owner_of(unsigned j, uint64_t i) const {
        for (uint32_t o = 0; o <= world; o++)
            if (F[j][o].has(i)) return (int)o;
        return -1;
    }
};

// What a rank's device does (prover.hip: GkrDevice) -- or, in the host self-test, plain loops.
// All indices are GLOBAL; the implementation knows where its window of each array lives.
struct GkrDeviceOps {
    virtual ~GkrDeviceOps() {}
    // first and last fraction of my range at every layer (entries of empty ranges are ignored)
    virtual void read_layer_edges(std::vector<std::array<Frac, 2>> &out) = 0;
    // a layer sumcheck begins: point = the layer's eq point (round entries)
    virtual void begin_layer(unsigned round, const std::vector<EF> &point) = 0;
    // Sub-round sr of layer `round + 1` on my data. table(sr) = the layer's records folded sr times
    // (sr == 0: the tree layer itself). If sr > 0, table(sr) is produced from table(sr - 1) with r_prev
    // for the records of `pairs` (2y, 2y+1 for y in pairs) and for the `n_edges` records edge_idx.
    // S[0..n_points) = sum over y in pairs of eq_rest(y) * pair(table(sr)[2y], table(sr)[2y+1])(X = 1..);
    // edges[k] = table(sr)[edge_idx[k]]; the indices are at most two runs of consecutive records (two unpaired
    // ends, or ONE run = the rank's whole remaining table handed to the host, at most GKR_HOST_TAIL records).
    // Blocks until the results are on the host.
    virtual void subround(unsigned round, unsigned sr, URange pairs, const uint64_t *edge_idx, unsigned n_edges, EF r_prev,
                          EF lambda, unsigned n_points, EF *S, GkrRec *edges) = 0;
};

// A rank whose table has shrunk to this many records hands all of them to the host in one go, and the host
// (every rank's, identically) finishes the layer: the last sub-rounds of a layer are a kernel launch and a
// mailbox round trip (~12 us) for a handful of pairs each, which the host sums in a microsecond or two.
static constexpr unsigned GKR_HOST_TAIL = 64;

struct GkrOutput {
    EF q0_claim = EF::one();
    std::vector<std::array<EF, 4>> claims;  // p0, p1, q0, q1 per layer
    std::vector<std::vector<std::array<EF, 3>>> polys;
    std::vector<EF> xi;
};

// The driver. `sample` / `observe` are the transcript; every rank calls this with the same plan and
// gets the same output.
struct GkrSharded {
    const GkrPlan &plan;
    const ShardComm &comm;
    GkrDeviceOps &dev;
    EF alpha;
    bool force_three_points = false;
    std::function<void(const EF &)> observe;
    std::function<EF()> sample;

    // host copies: edge fractions of every rank and layer, boundary fractions per layer
    std::vector<std::vector<std::array<Frac, 2>>> edges;  // [rank][layer] = {first, last}
    std::vector<std::map<uint64_t, Frac>> boundary;       // [layer]
    std::vector<EF> pad_q;                                // pad_q[j] = alpha^(2^(TR-j)): q of a padding-only node of layer j

    GkrSharded(const GkrPlan &p, const ShardComm &c, GkrDeviceOps &d, EF alpha_) : plan(p), comm(c), dev(d), alpha(alpha_) {}

    Frac frac_at(unsigned j, uint64_t i) const {
        const int o = plan.owner_of(j, i);
        if (o < 0) {
            auto it = boundary[j].find(i);
            if (it == boundary[j].end()) throw OvmFail(1, "internal: GKR boundary fraction missing");
            return it->second;
        }
        if ((uint32_t)o == plan.world) return Frac{EF::zero(), pad_q[j]};
        const URange &r = plan.F[j][o];
        if (i == r.lo) return edges[o][j][0];
        if (i == r.hi - 1) return edges[o][j][1];
        throw OvmFail(1, "internal: GKR fraction is interior to a rank");
    }

    void exchange_tree_edges() {
        const unsigned TR = plan.TR;
        std::vector<std::array<Frac, 2>> mine(TR + 1);
        for (auto &e : mine) e = {Frac{EF::zero(), EF::zero()}, Frac{EF::zero(), EF::zero()}};
        dev.read_layer_edges(mine);
        const size_t bytes = mine.size() * sizeof(mine[0]);
        std::vector<uint8_t> all(bytes * comm.world);
        comm.allgather(mine.data(), all.data(), bytes);
        edges.assign(comm.world, std::vector<std::array<Frac, 2>>(TR + 1));
        for (uint32_t r = 0; r < comm.world; r++) memcpy((void *)edges[r].data(), all.data() + r * bytes, bytes);
        pad_q.assign(TR + 1, alpha);
        for (unsigned j = TR; j-- > 0;) pad_q[j] = pad_q[j + 1] * pad_q[j + 1];
        // boundary fractions bottom-up: a fraction nobody owns whole is the sum of its children
        boundary.assign(TR + 1, {});
        for (unsigned j = TR; j > 0; j--) {
            // candidates: parents of the range ends and of the boundary fractions one layer down
            std::vector<uint64_t> cand;
            for (uint32_t o = 0; o < plan.owners(); o++) {
                const URange &r = plan.F[j][o];
                if (r.empty()) continue;
                cand.push_back(r.lo >> 1);
                cand.push_back((r.hi - 1) >> 1);
            }
            for (auto &kv : boundary[j]) cand.push_back(kv.first >> 1);
            for (uint64_t i : cand) {
                if (plan.owner_of(j - 1, i) >= 0 || boundary[j - 1].count(i)) continue;
                boundary[j - 1][i] = frac_add(frac_at(j, 2 * i), frac_at(j, 2 * i + 1));
            }
        }
    }

    GkrOutput run() {
        const unsigned TR = plan.TR;
        const uint32_t W = plan.world, me = comm.rank;
        GkrOutput out;
        exchange_tree_edges();
        {
            const Frac root = frac_at(0, 0);
            if (!is_zero(root.p)) throw OvmFail(1, "LogUp: non-zero root sum (interactions do not balance)");
            out.q0_claim = root.q;
            observe(out.q0_claim);
            const Frac l = frac_at(1, 0), r = frac_at(1, 1);
            out.claims.push_back({l.p, r.p, l.q, r.q});
        }
        auto observe_claims = [&](const std::array<EF, 4> &c) {
            observe(c[0]);
            observe(c[2]);
            observe(c[1]);
            observe(c[3]);
        };
        observe_claims(out.claims[0]);
        std::vector<EF> xi_prev{sample()};
        struct Msg {
            EF S[3];
            uint64_t n_edges;
            uint64_t eidx[GKR_HOST_TAIL];
            GkrRec edges[GKR_HOST_TAIL];
        };
        using HostRec = std::pair<uint64_t, GkrRec>;
        using HostRecs = std::vector<HostRec>;
        // scratch of the sub-round loop (kept across sub-rounds: no allocation per sub-round)
        HostRecs known;
        std::vector<Msg> all;
        std::vector<uint64_t> hy;
        std::vector<GkrRec> ha, hb, folded;
        std::vector<EF> he;
        std::vector<char> dumps;
        std::vector<URange> Pn;
        Msg mine{};
        const EF inv2 = EF::from_base(fp(2).inv());
        const bool trace_host = getenv("OVM_TRACE") != nullptr;  // (host time between two device sub-rounds, summed)
        auto host_t0 = std::chrono::steady_clock::now();
        uint64_t host_ns = 0, host_segments = 0;
        for (unsigned round = 1; round < TR; round++) {
            const unsigned L = round + 1;
            const EF lambda = sample();
            // record ranges of every owner; host records = the layer's boundary fractions' children
            std::vector<URange> R(plan.owners());
            for (uint32_t o = 0; o < plan.owners(); o++) R[o] = up(plan.F[L][o]);
            // (flat, sorted by record index: a layer's tail is six sub-rounds on 64, 32, ... records, and a tree node per
            // record per sub-round -- copy, two lookups, re-insert -- was half of the ~33 us the host spent per tail)
            HostRecs H;
            for (auto &kv : boundary[L - 1]) H.push_back({kv.first, rec_of(frac_at(L, 2 * kv.first), frac_at(L, 2 * kv.first + 1))});
            const GkrRec pad_rec{{EF::zero(), pad_q[L], EF::zero(), pad_q[L]}};
            dev.begin_layer(round, xi_prev);
            std::vector<std::array<EF, 3>> polys;
            std::vector<EF> rho;
            EF r_prev = EF::zero(), c_bound = EF::one();
            const auto &pc = out.claims.back();
            EF claim = (pc[0] + (pc[1] - pc[0]) * xi_prev[0]) + lambda * (pc[2] + (pc[3] - pc[2]) * xi_prev[0]);
            // sr == round is the epilogue: no sums, only the single record left (the layer's claims)
            GkrRec final_rec{};
            for (unsigned sr = 0; sr <= round; sr++) {
                const bool epilogue = sr == round;
                const unsigned nv = epilogue ? 0 : round - sr - 1;  // variables of eq_rest
                const EF *pt = xi_prev.data() + sr + 1;
                const EF a = epilogue ? EF::zero() : xi_prev[sr];
                const EF w0 = c_bound * (EF::one() - a);
                const unsigned n_points = epilogue ? 0 : ((is_zero(w0) || force_three_points) ? 3 : 2);
                // a rank with a short table hands ALL its records over and owns nothing from here on
                dumps.assign(W, 0);
                for (uint32_t o = 0; o < W; o++) dumps[o] = !R[o].empty() && R[o].size() <= GKR_HOST_TAIL;
                mine.n_edges = 0;  // (eidx / edges beyond n_edges are never read: no 4.6 KB clear per sub-round)
                for (auto &s : mine.S) s = EF::zero();
                URange my_records = R[me];
                for (uint32_t o = 0; o < W; o++)
                    if (dumps[o]) R[o] = URange{0, 0};
                Pn.resize(plan.owners());
                for (uint32_t o = 0; o < plan.owners(); o++) Pn[o] = epilogue ? URange{0, 0} : up(R[o]);
                // my device part
                if (dumps[me]) {
                    mine.n_edges = my_records.size();
                    for (uint64_t k = 0; k < mine.n_edges; k++) mine.eidx[k] = my_records.lo + k;
                } else {
                    mine.n_edges = range_edges(R[me], mine.eidx);
                }
                if (!Pn[me].empty() || mine.n_edges) {
                    if (trace_host) {
                        host_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - host_t0).count();
                        host_segments++;
                    }
                    dev.subround(round, sr, Pn[me], mine.eidx, (unsigned)mine.n_edges, r_prev, lambda, n_points, mine.S, mine.edges);
                    if (trace_host) host_t0 = std::chrono::steady_clock::now();
                }
                all.resize(W);
                if (W == 1) all[0] = mine;  // (allgather of one rank)
                else comm.allgather(&mine, all.data(), sizeof(Msg));
                // table records known to the host in this sub-round: edges of every owner + host records
                known.assign(H.begin(), H.end());
                EF S[3] = {EF::zero(), EF::zero(), EF::zero()};
                for (uint32_t o = 0; o < W; o++) {
                    for (unsigned k = 0; k < n_points; k++) S[k] += all[o].S[k];
                    for (uint64_t k = 0; k < all[o].n_edges && k < GKR_HOST_TAIL; k++) known.push_back({all[o].eidx[k], all[o].edges[k]});
                }
                {  // the padding run: constant records, closed-form sums
                    uint64_t pe[2];
                    const unsigned ne = range_edges(R[W], pe);
                    for (unsigned k = 0; k < ne; k++) known.push_back({pe[k], pad_rec});
                    if (!Pn[W].empty()) {
                        const EF wsum = eq_prefix_sum(pt, nv, Pn[W].hi) - eq_prefix_sum(pt, nv, Pn[W].lo);
                        rec_pair_sums(pad_rec, pad_rec, wsum, lambda, n_points, S);
                    }
                }
                // in index order; a record reported twice keeps the later report (what operator[] of a map did)
                std::stable_sort(known.begin(), known.end(), [](const HostRec &a, const HostRec &b) { return a.first < b.first; });
                {
                    size_t n = 0;
                    for (size_t i = 0; i < known.size(); i++) {
                        if (n && known[n - 1].first == known[i].first) known[n - 1] = known[i];
                        else known[n++] = known[i];
                    }
                    known.resize(n);
                }
                if (epilogue) {
                    if (known.empty() || known[0].first != 0) throw OvmFail(1, "internal: GKR layer claims missing");
                    final_rec = known[0].second;
                    break;
                }
                // pairs nobody owns: both records are known. Gathered first, then summed as ONE batch (eight pairs per
                // AVX2 step): a layer's tail is up to 32 + 16 + ... pairs, ~950 extension-field products
                hy.clear(), ha.clear(), hb.clear();
                for (size_t i = 0; i < known.size(); i += 2) {
                    const uint64_t y = known[i].first >> 1;
                    bool owned = false;
                    for (uint32_t o = 0; o < plan.owners(); o++) owned |= Pn[o].has(y);
                    if (owned) throw OvmFail(1, "internal: GKR edge record inside an owned pair");
                    // sorted and distinct: the two records of pair y are neighbours
                    if (i + 1 >= known.size() || known[i].first != 2 * y || known[i + 1].first != 2 * y + 1)
                        throw OvmFail(1, "internal: GKR host pair incomplete");
                    hy.push_back(y);
                    ha.push_back(known[i].second);
                    hb.push_back(known[i + 1].second);
                }
                if (!hy.empty()) {
                    he.resize(hy.size());
                    gkr_eq_batch(pt, nv, hy.data(), hy.size(), he.data());
                    gkr_pair_sums_batch(ha.data(), hb.data(), he.data(), hy.size(), lambda, n_points, S);
                }
                // s(X) = c_bound * eq1(a, X) * S(X), S quadratic: two points + the claim give the third
                EF Sv[3] = {S[0], S[1], S[2]};
                const EF a2 = a.dbl(), a3 = a2 + a, a5 = a3 + a2;
                const EF w[3] = {c_bound * a, c_bound * (a3 - EF::one()), c_bound * (a5 - EF::one().dbl())};
                EF S0;
                if (n_points == 2) {
                    S0 = (claim - w[0] * Sv[0]) * w0.inv();
                    Sv[2] = S0 - (Sv[0].dbl() + Sv[0]) + (Sv[1].dbl() + Sv[1]);
                } else {
                    S0 = (Sv[0].dbl() + Sv[0]) - (Sv[1].dbl() + Sv[1]) + Sv[2];
                }
                std::array<EF, 3> sv = {Sv[0] * w[0], Sv[1] * w[1], Sv[2] * w[2]};
                for (auto &e : sv) observe(e);
                polys.push_back(sv);
                r_prev = sample();
                rho.push_back(r_prev);
                c_bound = c_bound * (a * r_prev + (EF::one() - a) * (EF::one() - r_prev));
                const EF rm1 = r_prev - EF::one(), rm2 = rm1 - EF::one();
                claim = c_bound * (S0 * (rm1 * rm2) * inv2 - Sv[0] * (r_prev * rm2) + Sv[1] * (r_prev * rm1) * inv2);
                H.clear();
                if (!hy.empty()) {
                    folded.resize(hy.size());
                    gkr_fold_batch(ha.data(), hb.data(), hy.size(), r_prev, folded.data());
                    for (size_t i = 0; i < hy.size(); i++) H.push_back({hy[i], folded[i]});  // (ascending y: stays sorted)
                }
                R = Pn;
            }
            out.claims.push_back({final_rec.v[0], final_rec.v[2], final_rec.v[1], final_rec.v[3]});
            observe_claims(out.claims.back());
            const EF mu = sample();
            xi_prev.clear();
            xi_prev.push_back(mu);
            xi_prev.insert(xi_prev.end(), rho.begin(), rho.end());
            out.polys.push_back(std::move(polys));
        }
        out.xi = xi_prev;
        if (trace_host)
            fprintf(stderr, "[ovm] gkr host time between device sub-rounds: %.3f ms over %llu segments\n", host_ns / 1e6,
                    (unsigned long long)host_segments);
        return out;
    }
};

}  // namespace ovm
