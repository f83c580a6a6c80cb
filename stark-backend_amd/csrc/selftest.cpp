// Host-only self-tests of the sharded prover's bookkeeping (no device): the LogUp-GKR driver of
// gkr_shard.h run by `world` simulated ranks (threads, the shared-memory exchange between them) over
// plain host arrays, against the same driver on ONE rank that owns the whole tree with the padding
// leaves written out -- i.e. against the unsharded algorithm -- and the windowed eq-heap plan
// (kernels.h: EqHeapPlan) replayed on the host. What the device kernels compute is covered by the
// GPU parity tests; what is checked here is everything around them: who owns which fraction, which
// records straddle ranks, the closed-form padding sums, the exchange.
#include <thread>

#include <atomic>
#include <cstdlib>

#include "block_pool.h"
#include "gkr_shard.h"
#include "kernels.h"

using namespace ovm;

namespace {

struct Rng {
    uint64_t st;
    explicit Rng(uint64_t seed) : st(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint64_t next() {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Fp fp() { return Fp::from_canonical((uint32_t)(next() % bb::P)); }
    EF ef() {
        EF e;
        for (int k = 0; k < 4; k++) e.c[k] = fp();
        return e;
    }
};

// A rank's data as plain arrays: its window of every tree layer, and the folded tables as maps.
struct HostGkr : GkrDeviceOps {
    const GkrPlan &plan;
    uint32_t me;
    std::vector<std::vector<Frac>> layer;  // layer[j][i - F[j][me].lo]
    std::vector<EF> point;
    std::map<uint64_t, GkrRec> prev, cur;  // table(sr-1), table(sr): the records this rank materialised
    HostGkr(const GkrPlan &p, uint32_t me_, const std::vector<Frac> &all_leaves) : plan(p), me(me_), layer(p.TR + 1) {
        const unsigned TR = plan.TR;
        const URange &r = plan.F[TR][me];
        for (uint64_t i = r.lo; i < r.hi; i++) layer[TR].push_back(all_leaves[i]);
        for (unsigned j = TR; j-- > 0;) {
            const URange &pr = plan.F[j][me], &cr = plan.F[j + 1][me];
            for (uint64_t i = pr.lo; i < pr.hi; i++) layer[j].push_back(frac_add(layer[j + 1][2 * i - cr.lo], layer[j + 1][2 * i + 1 - cr.lo]));
        }
    }
    void read_layer_edges(std::vector<std::array<Frac, 2>> &out) override {
        for (unsigned j = 0; j <= plan.TR; j++)
            if (!layer[j].empty()) out[j] = {layer[j].front(), layer[j].back()};
    }
    void begin_layer(unsigned, const std::vector<EF> &pt) override {
        point = pt;
        prev.clear();
        cur.clear();
    }
    GkrRec table0(unsigned round, uint64_t y) const {
        const URange &r = plan.F[round + 1][me];
        if (!r.has(2 * y) || !r.has(2 * y + 1)) throw OvmFail(1, "selftest: record outside the rank's window");
        return rec_of(layer[round + 1][2 * y - r.lo], layer[round + 1][2 * y + 1 - r.lo]);
    }
    void subround(unsigned round, unsigned sr, URange pairs, const uint64_t *edge_idx, unsigned n_edges, EF r_prev, EF lambda,
                  unsigned n_points, EF *S, GkrRec *edges) override {
        auto source = [&](uint64_t k) -> GkrRec {  // record k of table(sr - 1)
            if (sr == 1) return table0(round, k);
            auto it = prev.find(k);
            if (it == prev.end()) throw OvmFail(1, "selftest: folded record was never materialised");
            return it->second;
        };
        auto record = [&](uint64_t k) -> GkrRec {  // record k of table(sr)
            if (sr == 0) return table0(round, k);
            return rec_fold(source(2 * k), source(2 * k + 1), r_prev);
        };
        cur.clear();
        const unsigned nv = sr < round ? round - sr - 1 : 0;
        for (uint64_t y = pairs.lo; y < pairs.hi; y++) {
            const GkrRec a = record(2 * y), b = record(2 * y + 1);
            cur[2 * y] = a;
            cur[2 * y + 1] = b;
            rec_pair_sums(a, b, eq_at(point.data() + sr + 1, nv, y), lambda, n_points, S);
        }
        for (unsigned k = 0; k < n_edges; k++) edges[k] = record(edge_idx[k]);
        if (sr > 0) prev = cur;  // (the device keeps exactly these: the records of its owned pairs)
        else prev.clear();
    }
};

struct RunResult {
    GkrOutput out;
    std::array<uint32_t, 16> state;
    std::string error;
};

RunResult run_rank(const GkrPlan &plan, const ShardComm &comm, const std::vector<Frac> &leaves, EF alpha, bool three) {
    RunResult res;
    try {
        HostGkr dev(plan, comm.rank, leaves);
        p2::Sponge sp;
        sp.init();
        GkrSharded drv(plan, comm, dev, alpha);
        drv.force_three_points = three;
        drv.observe = [&](const EF &e) { sp.observe_ext(e); };
        drv.sample = [&]() { return sp.sample_ext(); };
        res.out = drv.run();
        for (int i = 0; i < 16; i++) res.state[i] = sp.state[i].v;
    } catch (const std::exception &e) {
        res.error = e.what();
    }
    return res;
}

bool same(const RunResult &a, const RunResult &b) {
    if (!a.error.empty() || !b.error.empty()) return false;
    if (a.state != b.state || !(a.out.q0_claim == b.out.q0_claim) || a.out.xi.size() != b.out.xi.size()) return false;
    for (size_t i = 0; i < a.out.xi.size(); i++)
        if (!(a.out.xi[i] == b.out.xi[i])) return false;
    if (a.out.claims.size() != b.out.claims.size() || a.out.polys.size() != b.out.polys.size()) return false;
    for (size_t i = 0; i < a.out.claims.size(); i++)
        for (int k = 0; k < 4; k++)
            if (!(a.out.claims[i][k] == b.out.claims[i][k])) return false;
    for (size_t i = 0; i < a.out.polys.size(); i++) {
        if (a.out.polys[i].size() != b.out.polys[i].size()) return false;
        for (size_t j = 0; j < a.out.polys[i].size(); j++)
            for (int k = 0; k < 3; k++)
                if (!(a.out.polys[i][j][k] == b.out.polys[i][j][k])) return false;
    }
    return true;
}

// Replays an EqHeapPlan on the host (what eq_heap_top_kernel / eq_heap_levels_win_kernel write) and
// checks every needed entry against the definition.
void check_heap_plan(uint32_t n, const std::vector<std::pair<uint64_t, uint64_t>> &need, Rng &rng) {
    const EqHeapPlan pl = eq_heap_plan(n, need);
    bool any = false;
    for (auto &w : need) any |= w.first < w.second;
    if (pl.empty != !any) throw OvmFail(1, "selftest: eq heap plan emptiness");
    if (pl.empty) return;
    std::vector<EF> point(n);
    for (auto &x : point) x = rng.ef();
    std::vector<EF> heap(pl.total, EF::zero());
    std::vector<char> written(pl.total, 0);
    for (uint32_t d = 0; d <= pl.top; d++)
        for (uint64_t y = 0; y < (uint64_t(1) << d); y++) {
            heap[(uint64_t(1) << d) + y] = eq_at(point.data() + (n - d), d, y);
            written[(uint64_t(1) << d) + y] = 1;
        }
    for (auto &ln : pl.launches)
        for (uint64_t t = 0; t < ln.count; t++) {
            const uint64_t i = ln.src_first + t;
            const uint64_t src = pl.at(ln.k, i);
            if (src >= pl.total || !written[src]) throw OvmFail(1, "selftest: eq heap launch reads an entry nobody wrote");
            std::vector<EF> cur{heap[src]};
            for (uint32_t l = 0; l < ln.levels; l++) {
                const EF x = point[n - 1 - (ln.k + l)];
                std::vector<EF> nxt(cur.size() * 2);
                for (size_t u = 0; u < cur.size(); u++) {
                    nxt[2 * u + 1] = cur[u] * x;
                    nxt[2 * u] = cur[u] - nxt[2 * u + 1];
                }
                cur = nxt;
                for (size_t u = 0; u < cur.size(); u++) {
                    const uint64_t y = (i << (l + 1)) + u;
                    const uint64_t at = pl.at(ln.k + l + 1, y);
                    if (y < pl.lo[ln.k + l + 1] || y >= pl.lo[ln.k + l + 1] + pl.cnt[ln.k + l + 1] || at >= pl.total)
                        throw OvmFail(1, "selftest: eq heap launch writes outside the stored window");
                    heap[at] = cur[u];
                    written[at] = 1;
                }
            }
        }
    for (uint32_t d = 0; d < n; d++)
        for (uint64_t y = need[d].first; y < need[d].second; y++) {
            const uint64_t at = pl.at(d, y);
            if (y < pl.lo[d] || y >= pl.lo[d] + pl.cnt[d] || at >= pl.total || !written[at])
                throw OvmFail(1, "selftest: a needed eq weight is not stored");
            if (!(heap[at] == eq_at(point.data() + (n - d), d, y))) throw OvmFail(1, "selftest: eq weight differs");
        }
}

thread_local std::string g_selftest_error;

}  // namespace

extern "C" {

// `world` ranks own the leaf runs [0, ends[0]), [ends[0], ends[1]), ... ; leaves [ends[world-1], 2^TR) are padding.
// 0 = every rank's transcript equals the unsharded run's. Needs no device.
int ovm_hip_gkr_shard_selftest(uint32_t world, uint32_t TR, const uint64_t *ends, uint64_t seed, int three_points) {
    try {
        if (world == 0 || world > 64 || TR == 0 || TR > 20 || !ends) return 1;
        const uint64_t n_leaves = uint64_t(1) << TR;
        Rng rng(seed);
        const EF alpha = rng.ef();
        std::vector<URange> ranges(world);
        uint64_t at = 0;
        for (uint32_t r = 0; r < world; r++) {
            if (ends[r] < at || ends[r] > n_leaves) return 1;
            ranges[r] = ends[r] > at ? URange{at, ends[r]} : URange{0, 0};
            at = ends[r];
        }
        const uint64_t used = at;
        // leaves: pairs (p, q), (-p, q) so that the fractions sum to zero, padding (0, alpha)
        std::vector<Frac> leaves(n_leaves, Frac{EF::zero(), alpha});
        for (uint64_t i = 0; i + 1 < used; i += 2) {
            const EF p = EF::from_base(rng.fp()), q = rng.ef();
            leaves[i] = Frac{p, q};
            leaves[i + 1] = Frac{EF::zero() - p, q};
        }
        if (used & 1) leaves[used - 1] = Frac{EF::zero(), rng.ef()};
        // reference: one rank, the padding written out
        const GkrPlan ref_plan(TR, {URange{0, n_leaves}}, n_leaves);
        const RunResult ref = run_rank(ref_plan, ShardComm(), leaves, alpha, three_points != 0);
        if (!ref.error.empty()) throw OvmFail(1, "selftest: reference run failed: " + ref.error);
        // sharded: threads over the shared-memory exchange
        const GkrPlan plan(TR, ranges, used);
        const uint64_t slot = 4096;
        std::vector<uint8_t> seg(ovm_hip_shm_segment_bytes(world, slot) + 64, 0);
        uint8_t *base = (uint8_t *)(((uintptr_t)seg.data() + 63) & ~uintptr_t(63));
        std::vector<RunResult> res(world);
        std::vector<std::thread> th;
        for (uint32_t r = 0; r < world; r++)
            th.emplace_back([&, r] {
                ovm_shm_comm_t *c = nullptr;
                if (ovm_hip_shm_comm_attach(base, r, world, slot, &c) != 0) {
                    res[r].error = "attach failed";
                    return;
                }
                ovm_shard_t abi{r, world, c, ovm_hip_shm_allgather, nullptr};
                res[r] = run_rank(plan, ShardComm(&abi), leaves, alpha, three_points != 0);
                ovm_hip_shm_comm_destroy(c);
            });
        for (auto &t : th) t.join();
        for (uint32_t r = 0; r < world; r++) {
            if (!res[r].error.empty()) throw OvmFail(1, "selftest: rank " + std::to_string(r) + ": " + res[r].error);
            if (!same(res[r], ref)) throw OvmFail(1, "selftest: rank " + std::to_string(r) + " differs from the unsharded run");
        }
        // the eq-heap windows each rank would ask for, replayed
        for (uint32_t r = 0; r < world; r++)
            for (unsigned round = 1; round < TR; round++) {
                std::vector<URange> c(round + 2);
                c[0] = up(plan.F[round + 1][r]);
                for (unsigned s2 = 0; s2 <= round; s2++) c[s2 + 1] = up(c[s2]);
                std::vector<std::pair<uint64_t, uint64_t>> need(round);
                for (unsigned d = 0; d < round; d++) need[d] = {c[round - d].lo, c[round - d].hi};
                check_heap_plan(round, need, rng);
            }
        return 0;
    } catch (const std::exception &e) {
        g_selftest_error = e.what();
        return 1;
    }
}
const char *ovm_hip_selftest_error(void) { return g_selftest_error.c_str(); }

// The bulk exchange of a sharded proof on HOST buffers (for ranks without a GPU: the CPU test of the callback
// the host language supplies): every rank fills a vector of n extension-field elements from (seed, rank),
// all-gathers the vectors through shard->allgather_device -- which here receives host pointers and a null
// stream -- and adds the blocks up with the field's own addition; the result must equal the sum of all ranks'
// vectors, which every rank can regenerate. Also runs `rounds` small exchanges through allgather_host whose
// contents depend on the round (order, sizes, chunking). 0 = all as expected.
int ovm_hip_shard_exchange_selftest(const ovm_shard_t *shard, uint64_t n, uint32_t rounds, uint64_t seed) {
    try {
        if (!shard || shard->world == 0 || shard->rank >= shard->world) return 1;
        const ShardComm comm(shard);
        const uint32_t W = shard->world;
        auto vec_of = [&](uint32_t rank) {
            Rng rng(seed * 1315423911ull + rank);
            std::vector<EF> v(n);
            for (auto &x : v) x = rng.ef();
            return v;
        };
        if (W > 1 && n) {
            const std::vector<EF> mine = vec_of(shard->rank);
            std::vector<EF> all(n * W);
            comm.allgather_device(mine.data(), all.data(), n * sizeof(EF), nullptr);
            std::vector<EF> sum(n, EF::zero());
            for (uint32_t r = 0; r < W; r++) {
                const std::vector<EF> v = vec_of(r);
                for (uint64_t i = 0; i < n; i++) {
                    if (!(all[r * n + i] == v[i])) throw OvmFail(1, "selftest: block " + std::to_string(r) + " of the bulk exchange differs");
                    sum[i] += v[i];
                }
            }
            for (uint64_t i = 0; i < n; i++) {
                EF acc = all[i];
                for (uint32_t r = 1; r < W; r++) acc = acc + all[r * n + i];
                if (!(acc == sum[i])) throw OvmFail(1, "selftest: sum of the gathered blocks differs");
            }
            if (comm.has_alltoall()) {
                // the same sum as the prover forms it when the all-to-all is there: slices of ceil(n / W)
                const uint64_t slice = (n + W - 1) / W;
                std::vector<EF> send(slice * W, EF::zero()), recv(slice * W), red(slice), full(slice * W);
                std::copy(mine.begin(), mine.end(), send.begin());
                comm.alltoall_device(send.data(), recv.data(), slice * sizeof(EF), nullptr);
                for (uint64_t i = 0; i < slice; i++) {
                    EF acc = recv[i];
                    for (uint32_t r = 1; r < W; r++) acc = acc + recv[r * slice + i];
                    red[i] = acc;
                }
                comm.allgather_device(red.data(), full.data(), slice * sizeof(EF), nullptr);
                for (uint64_t i = 0; i < n; i++)
                    if (!(full[i] == sum[i])) throw OvmFail(1, "selftest: all-to-all form of the sum differs at " + std::to_string(i));
            }
        }
        for (uint32_t k = 0; k < rounds; k++) {
            // sizes differ per round and rank (allgatherv), some beyond one slot
            Rng rng(seed + 77 * k);
            std::vector<uint64_t> sizes(W);
            for (auto &sz : sizes) sz = rng.next() % (k % 16 == 0 ? 300000 : 400);
            auto payload = [&](uint32_t rank) {
                std::vector<uint8_t> b(sizes[rank]);
                Rng g(seed ^ (uint64_t(k) << 20) ^ rank);
                for (auto &x : b) x = (uint8_t)g.next();
                return b;
            };
            const auto got = comm.allgatherv(payload(shard->rank));
            for (uint32_t r = 0; r < W; r++)
                if (got[r] != payload(r)) throw OvmFail(1, "selftest: small exchange " + std::to_string(k) + " differs");
        }
        return 0;
    } catch (const std::exception &e) {
        g_selftest_error = e.what();
        return 1;
    }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// The device memory pool (block_pool.h) over plain host memory: a random allocate / free workload of
// both size classes, every block filled with its own pattern, with the invariants checked as it goes --
// blocks never overlap, contents survive until the free, what the pool holds from the "driver" is what
// is live plus what it reports as free, a limit is honoured as soon as whole segments are free, a
// second identical pass makes no driver call, everything goes back at the end. 0 = all held.
// ---------------------------------------------------------------------------------------------
namespace {
std::atomic<size_t> g_host_driver_bytes{0};
std::atomic<uint64_t> g_host_driver_allocs{0};
std::mutex g_host_driver_mu;
std::unordered_map<void *, size_t> g_host_driver_sizes;
PoolDriver host_pool_driver() {
    return PoolDriver{[](void **p, size_t bytes) {
                          if (posix_memalign(p, 4096, bytes) != 0) return 2;  // (hipMalloc aligns at least this much)
                          std::lock_guard<std::mutex> lk(g_host_driver_mu);
                          g_host_driver_sizes[*p] = bytes;
                          g_host_driver_bytes += bytes;
                          g_host_driver_allocs++;
                          return 0;
                      },
                      [](int, void *p) {
                          std::lock_guard<std::mutex> lk(g_host_driver_mu);
                          g_host_driver_bytes -= g_host_driver_sizes[p];
                          g_host_driver_sizes.erase(p);
                          std::free(p);
                      },
                      [](int) {}, 0};
}
}  // namespace

extern "C" int ovm_hip_pool_selftest(uint64_t seed, uint32_t ops) {
    try {
        Rng rng(seed);
        BlockPool *pool = new BlockPool(host_pool_driver());
        struct Live {
            char *p;
            size_t want;
            uint8_t tag;
        };
        std::vector<Live> live;
        std::map<char *, size_t> spans;  // start -> requested size, for the overlap check
        size_t live_want = 0;
        auto fail = [](const std::string &m) { throw OvmFail(1, "pool selftest: " + m); };
        auto alloc = [&](size_t want) {
            char *p = (char *)pool->get(want);
            if ((uintptr_t)p % 512) fail("block not 512-byte aligned");
            auto nx = spans.lower_bound(p);
            if (nx != spans.end() && nx->first < p + want) fail("block overlaps the next live block");
            if (nx != spans.begin()) {
                auto pv = std::prev(nx);
                if (pv->first + pv->second > p) fail("block overlaps the previous live block");
            }
            const uint8_t tag = (uint8_t)rng.next();
            p[0] = (char)tag, p[want / 2] = (char)(tag ^ 0x5a), p[want - 1] = (char)(tag ^ 0xa5);
            spans[p] = want;
            live.push_back({p, want, tag});
            live_want += want;
        };
        auto release = [&](size_t i) {
            Live l = live[i];
            if ((uint8_t)l.p[0] != l.tag || (uint8_t)l.p[l.want / 2] != (uint8_t)(l.tag ^ 0x5a) || (uint8_t)l.p[l.want - 1] != (uint8_t)(l.tag ^ 0xa5))
                fail("block contents changed while it was live");
            spans.erase(l.p);
            live_want -= l.want;
            live[i] = live.back();
            live.pop_back();
            BlockPool::put(l.p, pool);
        };
        auto size_of = [&]() -> size_t {
            const uint64_t r = rng.next() % 100;
            if (r < 55) return 1 + rng.next() % (BlockPool::SMALL_MAX - 1);        // small class
            if (r < 95) return BlockPool::SMALL_MAX + rng.next() % (size_t(24) << 20);  // 1-25 MiB
            return (size_t(32) << 20) + rng.next() % (size_t(96) << 20);            // a few big ones
        };
        for (uint32_t k = 0; k < ops; k++) {
            if (live.empty() || (rng.next() % 100 < 55 && live_want < (size_t(1) << 30)))
                alloc(size_of());
            else
                release(rng.next() % live.size());
            if (k % 64 == 0 && g_host_driver_bytes.load() < live_want + pool->cached_bytes() - pool->cached_bytes())
                fail("accounting: less held from the driver than is live");
            if (k % 257 == 0) {
                // everything the driver gave is either live (rounded up) or reported free
                if (g_host_driver_bytes.load() < live_want) fail("accounting: driver bytes below live bytes");
                if (pool->cached_bytes() > g_host_driver_bytes.load()) fail("accounting: more free than held");
            }
        }
        // a limit: with everything freed, at most `limit` bytes may stay
        while (!live.empty()) release(live.size() - 1);
        if (pool->cached_bytes() != g_host_driver_bytes.load())
            fail("all blocks free, but free bytes " + std::to_string(pool->cached_bytes()) + " != bytes held " + std::to_string(g_host_driver_bytes.load()));
        pool->set_limit(size_t(16) << 20);
        if (g_host_driver_bytes.load() > (size_t(16) << 20)) fail("limit not honoured with every segment free");
        pool->set_limit(0);
        // steady state: the same request sequence twice -- the second pass must not reach the driver
        auto pass = [&]() {
            Rng r2(seed ^ 0x1234);
            std::vector<size_t> keep;
            for (int k = 0; k < 300; k++) {
                const uint64_t r = r2.next() % 100;
                alloc(r < 50 ? 1 + r2.next() % 200000 : (size_t(1) << 20) + r2.next() % (size_t(40) << 20));
                if (r2.next() % 3 == 0 && !live.empty()) release(r2.next() % live.size());
            }
            while (!live.empty()) release(live.size() - 1);
        };
        // (a pass that had to grow the pool leaves more segments than it found, so the next one may choose other
        // blocks: the pool settles within a few passes; once a pass made no driver call, every later one is its replay)
        int settled_after = -1;
        for (int k = 0; k < 8 && settled_after < 0; k++) {
            const uint64_t before = g_host_driver_allocs.load();
            pass();
            if (g_host_driver_allocs.load() == before) settled_after = k;
        }
        if (settled_after < 0) fail("a repeated request sequence still reaches the driver after 8 passes");
        const uint64_t before = g_host_driver_allocs.load();
        pass();
        pass();
        if (g_host_driver_allocs.load() != before) fail("a settled request sequence reached the driver again");
        pool->trim();
        if (g_host_driver_bytes.load() != 0) fail("trim with nothing live left segments behind");
        if (pool->orphan()) delete pool;
        else fail("orphaned empty pool not reported dead");
        return 0;
    } catch (const std::exception &e) {
        g_selftest_error = e.what();
        return 1;
    }
}
