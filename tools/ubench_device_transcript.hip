// What would a DEVICE-RESIDENT TRANSCRIPT buy the small sub-rounds of the GKR layer sumchecks?
// Measured, not modelled: a synthetic layer sumcheck with the real arithmetic -- pair sums of fraction records with eq
// weights, the block reduction, the host algebra of a sub-round (gkr_shard.h: an extension-field inversion, the third
// evaluation, the claim update), a duplex sponge (rate 8: 12 words observed + 4 sampled = 2 or 3 permutations per sub-round),
// the fold by the sampled challenge -- run three ways on tables of 2^7 .. 2^10 records:
//   device     ONE persistent workgroup, the table in LDS, the sponge on 16 lanes (p2::permute_coop), the algebra on one
//              lane: no launch and no host round trip per sub-round                       [what item 3 asks for]
//   server     the same persistent workgroup, but sums go to the host through a mapped mailbox and the challenge comes back
//              through a mapped slot (the host runs the sponge: stand-in of HOST_WORK_US): no launch, one PCIe round trip
//   launches   one kernel launch per sub-round (table in HBM/L2) + mailbox + host work + next launch   [what the prover does]
// and prints microseconds per sub-round. Nothing here is protocol-exact (random records, stand-in weights); the per-sub-round
// WORK is the prover's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark-backend_amd/csrc -I include tools/ubench_device_transcript.hip -o tools/ubench_device_transcript
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bb31.h"
#include "poseidon2.h"

using bb::Fp;
using EF = bb::FpExt;
static constexpr double HOST_WORK_US = 2.5;  // the host's share of a sub-round in the prover (2-3 AVX2 permutations + algebra)

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

struct Rec {
    EF v[4];  // p0, q0, p1, q1
};

__device__ __forceinline__ void pair_sums(const Rec &a, const Rec &b, const EF &e, const EF &lambda, EF *s) {
    EF v[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = b.v[k];
        d[k] = b.v[k] - a.v[k];
    }
#pragma unroll
    for (int x = 0; x < 2; x++) {
        s[x] = s[x] + (v[0] * v[3] + v[1] * (v[2] + lambda * v[3])) * e;
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = v[k] + d[k];
    }
}
__device__ __forceinline__ Rec fold(const Rec &a, const Rec &b, const EF &r) {
    Rec o;
#pragma unroll
    for (int k = 0; k < 4; k++) o.v[k] = a.v[k] + (b.v[k] - a.v[k]) * r;
    return o;
}
// two sums over the workgroup -> sh_s[0..2) (every thread calls it; `scratch` holds 2 EF per wave)
__device__ __forceinline__ void block_sum2(EF *s, EF *scratch, EF *out) {
    for (int x = 0; x < 2; x++)
        for (int c = 0; c < 4; c++) {
            uint32_t w = s[x].c[c].v;
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_down(w, off);
                w = (Fp::raw(w) + Fp::raw(o)).v;
            }
            s[x].c[c].v = w;
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    if (lane == 0) {
        scratch[2 * wave] = s[0];
        scratch[2 * wave + 1] = s[1];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        EF a = scratch[0], b = scratch[1];
        for (int w2 = 1; w2 < n_waves; w2++) {
            a = a + scratch[2 * w2];
            b = b + scratch[2 * w2 + 1];
        }
        out[0] = a;
        out[1] = b;
    }
}
// the host algebra of one sub-round (gkr_shard.h:387-405), on one lane: S0 from the claim (an inversion), the third value,
// the three observed values; `claim` and `c_bound` updated with the challenge afterwards (finish_algebra)
struct Algebra {
    EF claim, c_bound, a, S0, Sv[3];
};
__device__ __forceinline__ void algebra(Algebra &g, const EF *S, EF *sv) {
    const EF one = EF::one();
    const EF w0 = g.c_bound * (one - g.a);
    const EF a2 = g.a + g.a, a3 = a2 + g.a, a5 = a3 + a2;
    const EF w[3] = {g.c_bound * g.a, g.c_bound * (a3 - one), g.c_bound * (a5 - (one + one))};
    g.Sv[0] = S[0];
    g.Sv[1] = S[1];
    g.S0 = (g.claim - w[0] * g.Sv[0]) * w0.inv();
    g.Sv[2] = g.S0 - (g.Sv[0] + g.Sv[0] + g.Sv[0]) + (g.Sv[1] + g.Sv[1] + g.Sv[1]);
    for (int k = 0; k < 3; k++) sv[k] = g.Sv[k] * w[k];
}
__device__ __forceinline__ void finish_algebra(Algebra &g, const EF &r, const EF &next_a) {
    const EF one = EF::one();
    g.c_bound = g.c_bound * (g.a * r + (one - g.a) * (one - r));
    const EF rm1 = r - one, rm2 = rm1 - one;
    const EF inv2 = EF::from_base(Fp::from_canonical(1006632961u));  // 1/2
    g.claim = g.c_bound * (g.S0 * (rm1 * rm2) * inv2 - g.Sv[0] * (r * rm2) + g.Sv[1] * (r * rm1) * inv2);
    g.a = next_a;
}

// ---- device: everything in one persistent workgroup -------------------------------------------------------------
// LDS: two tables of n records (ping-pong), 2 EF per wave of scratch, sums, observed values, the challenge.
__global__ void k_device(const Rec *table, const EF *eq, uint32_t log_n, uint32_t layers, uint32_t *out_state, long long *ticks) {
    extern __shared__ uint8_t lds[];
    const uint32_t n = 1u << log_n;
    Rec *t0 = (Rec *)lds, *t1 = t0 + n;
    EF *scratch = (EF *)(t1 + n), *S = scratch + 2 * (blockDim.x >> 6), *sv = S + 2, *r_sh = sv + 3;
    const uint32_t lane16 = threadIdx.x & 15;
    Fp sponge = Fp::from_canonical(threadIdx.x & 15);  // wave 0, lanes 0..15 (the other lanes compute along, unused)
    Algebra g{EF::one(), EF::one(), EF::from_base(Fp::from_canonical(7)), EF::zero(), {EF::zero(), EF::zero(), EF::zero()}};
    const EF lambda = EF::from_base(Fp::from_canonical(5));
    const long long c0 = wall_clock64();
    uint32_t round_ctr = 0;
    for (uint32_t layer = 0; layer < layers; layer++) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) t0[i] = table[i];
        __syncthreads();
        Rec *cur = t0, *nxt = t1;
        const EF *eq_l = eq;
        for (uint32_t sr = 0; sr < log_n; sr++, round_ctr++) {
            const uint32_t pairs = n >> (sr + 1);
            EF s[2] = {EF::zero(), EF::zero()};
            for (uint32_t y = threadIdx.x; y < pairs; y += blockDim.x) pair_sums(cur[2 * y], cur[2 * y + 1], eq_l[y], lambda, s);
            block_sum2(s, scratch, S);
            if (threadIdx.x == 0) algebra(g, S, sv);
            __syncthreads();
            if (threadIdx.x < 64) {
                // duplex sponge, rate 8: 12 observed words = one full block + 4 words; the sample after an observation
                // permutes again; every second sub-round a third permutation (the buffer positions drift by 4 per sub-round)
                const uint32_t *obs = (const uint32_t *)sv;
                if (lane16 < 8) sponge = Fp::raw(obs[lane16]);
                sponge = p2::permute_coop(sponge, lane16);
                if (lane16 < 4) sponge = Fp::raw(obs[8 + lane16]);
                sponge = p2::permute_coop(sponge, lane16);
                if (round_ctr & 1) sponge = p2::permute_coop(sponge, lane16);
                if (threadIdx.x < 4) r_sh->c[threadIdx.x] = sponge;
            }
            __syncthreads();
            const EF r = *r_sh;
            if (threadIdx.x == 0) finish_algebra(g, r, r);
            for (uint32_t y = threadIdx.x; y < pairs; y += blockDim.x) nxt[y] = fold(cur[2 * y], cur[2 * y + 1], r);
            __syncthreads();
            Rec *t = cur;
            cur = nxt;
            nxt = t;
            eq_l += pairs;
        }
    }
    if (threadIdx.x < 16) out_state[threadIdx.x] = sponge.v;
    if (threadIdx.x == 0) {
        out_state[16] = g.claim.c[0].v;
        ticks[0] = wall_clock64() - c0;
    }
}

// ---- server: the same workgroup, challenge from the host through mapped memory ------------------------------------
struct Mail {
    volatile uint32_t sums[12];
    volatile uint32_t seq;      // device -> host: sums of sub-round seq are there
    uint32_t pad[19];
    volatile uint32_t r[4];
    volatile uint32_t r_seq;    // host -> device: the challenge of sub-round r_seq
};
template <bool FENCED>
__global__ void k_server(const Rec *table, const EF *eq, uint32_t log_n, uint32_t layers, Mail *mail, uint32_t *out_state) {
    extern __shared__ uint8_t lds[];
    const uint32_t n = 1u << log_n;
    Rec *t0 = (Rec *)lds, *t1 = t0 + n;
    EF *scratch = (EF *)(t1 + n), *S = scratch + 2 * (blockDim.x >> 6), *r_sh = S + 2;
    const EF lambda = EF::from_base(Fp::from_canonical(5));
    uint32_t seq = 0;
    for (uint32_t layer = 0; layer < layers; layer++) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) t0[i] = table[i];
        __syncthreads();
        Rec *cur = t0, *nxt = t1;
        const EF *eq_l = eq;
        for (uint32_t sr = 0; sr < log_n; sr++) {
            const uint32_t pairs = n >> (sr + 1);
            EF s[2] = {EF::zero(), EF::zero()};
            for (uint32_t y = threadIdx.x; y < pairs; y += blockDim.x) pair_sums(cur[2 * y], cur[2 * y + 1], eq_l[y], lambda, s);
            block_sum2(s, scratch, S);
            seq++;
            if (threadIdx.x == 0) {
                const uint32_t *w = (const uint32_t *)S;
                for (int k = 0; k < 8; k++) __hip_atomic_store((uint32_t *)&mail->sums[k], w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                // (bounded polls: a host that went away must not leave a kernel spinning on the box)
                if (FENCED) {
                    __hip_atomic_store((uint32_t *)&mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    for (uint32_t spins = 0; __hip_atomic_load((uint32_t *)&mail->r_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq && spins < (1u << 22); spins++) {
                    }
                } else {
                    // the library's fence-free form (device_common.h): write-through stores, wait for their acknowledgement,
                    // then the flag; the poll and the challenge are system-scope loads, which never hit a cache
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store((uint32_t *)&mail->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    for (uint32_t spins = 0; __hip_atomic_load((uint32_t *)&mail->r_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq && spins < (1u << 22); spins++) {
                    }
                }
                for (int k = 0; k < 4; k++) r_sh->c[k].v = __hip_atomic_load((uint32_t *)&mail->r[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __syncthreads();
            const EF r = *r_sh;
            for (uint32_t y = threadIdx.x; y < pairs; y += blockDim.x) nxt[y] = fold(cur[2 * y], cur[2 * y + 1], r);
            __syncthreads();
            Rec *t = cur;
            cur = nxt;
            nxt = t;
            eq_l += pairs;
        }
    }
    if (threadIdx.x < 4) out_state[threadIdx.x] = r_sh->c[threadIdx.x].v;
}

// ---- launches: one kernel per sub-round, table in global memory (what the prover does for these sizes) ---------------
__global__ void k_round(const Rec *in, Rec *out, const EF *eq_l, uint32_t pairs, int do_fold, EF r, Mail *mail, uint32_t seq) {
    __shared__ EF scratch[2 * 16 + 2];
    const EF lambda = EF::from_base(Fp::from_canonical(5));
    EF s[2] = {EF::zero(), EF::zero()};
    // fold the previous table by r on the way in (the prover's fused fold + sums: reads 2 records per output record)
    for (uint32_t y = threadIdx.x; y < pairs; y += blockDim.x) {
        Rec a, b;
        if (do_fold) {
            a = fold(in[4 * y], in[4 * y + 1], r);
            b = fold(in[4 * y + 2], in[4 * y + 3], r);
            out[2 * y] = a;
            out[2 * y + 1] = b;
        } else {
            a = in[2 * y];
            b = in[2 * y + 1];
        }
        pair_sums(a, b, eq_l[y], lambda, s);
    }
    block_sum2(s, scratch, scratch + 32);
    if (threadIdx.x == 0) {
        const uint32_t *w = (const uint32_t *)(scratch + 32);
        for (int k = 0; k < 8; k++) __hip_atomic_store((uint32_t *)&mail->sums[k], w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the prover's fence-free mailbox)
        __hip_atomic_store((uint32_t *)&mail->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static void wait_seq(Mail *mail, uint32_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t polls = 0;
    while (__atomic_load_n((uint32_t *)&mail->seq, __ATOMIC_ACQUIRE) != s)
        if ((++polls & 0xffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
            fprintf(stderr, "no answer from the device for sub-round %u\n", s);
            exit(2);
        }
}
static void host_work() {
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < HOST_WORK_US) {
    }
}

int main() {
    const uint32_t max_log = 10, layers = 40;
    std::vector<Rec> h_table(1u << max_log);
    uint32_t x = 12345;
    for (auto &rec : h_table)
        for (auto &e : rec.v)
            for (auto &c : e.c) {
                x = x * 1664525u + 1013904223u;
                c = Fp::from_canonical(x % bb::P);
            }
    std::vector<EF> h_eq(1u << max_log);
    for (auto &e : h_eq)
        for (auto &c : e.c) {
            x = x * 1664525u + 1013904223u;
            c = Fp::from_canonical(x % bb::P);
        }
    Rec *d_table, *d_a, *d_b;
    EF *d_eq;
    uint32_t *d_state;
    long long *d_ticks;
    Mail *mail;
    CHECK(hipMalloc(&d_table, sizeof(Rec) << max_log));
    CHECK(hipMalloc(&d_a, sizeof(Rec) << max_log));
    CHECK(hipMalloc(&d_b, sizeof(Rec) << max_log));
    CHECK(hipMalloc(&d_eq, sizeof(EF) << max_log));
    CHECK(hipMalloc(&d_state, 64 * 4));
    CHECK(hipMalloc(&d_ticks, 64));
    CHECK(hipHostMalloc((void **)&mail, sizeof(Mail), hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipMemcpy(d_table, h_table.data(), sizeof(Rec) << max_log, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_eq, h_eq.data(), sizeof(EF) << max_log, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    CHECK(hipFuncSetAttribute((const void *)k_device, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CHECK(hipFuncSetAttribute((const void *)k_server<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CHECK(hipFuncSetAttribute((const void *)k_server<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    printf("records  threads   device us/sub-round          server (fences)   server (fence-free)   launches us/sub-round   (host work %.1f us in the last three)\n", HOST_WORK_US);
    for (uint32_t log_n = 7; log_n <= max_log; log_n++) {
        for (uint32_t threads : {256u, 1024u}) {
            const size_t lds = (sizeof(Rec) << (log_n + 1)) + 1024;
            const uint32_t rounds = layers * log_n;
            double us_dev = 0, us_srv = 0, us_srv_ff = 0, us_launch = 0;
            for (int rep = 0; rep < 3; rep++) {  // (the last repetition counts)
                // device
                auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k_device, dim3(1), dim3(threads), lds, st, d_table, d_eq, log_n, layers, d_state, d_ticks);
                CHECK(hipStreamSynchronize(st));
                us_dev = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
                // server, with fences and without
                for (int fenced = 1; fenced >= 0; fenced--) {
                    mail->seq = 0;
                    mail->r_seq = 0;
                    t0 = std::chrono::steady_clock::now();
                    if (fenced) hipLaunchKernelGGL(k_server<true>, dim3(1), dim3(threads), lds, st, d_table, d_eq, log_n, layers, mail, d_state);
                    else hipLaunchKernelGGL(k_server<false>, dim3(1), dim3(threads), lds, st, d_table, d_eq, log_n, layers, mail, d_state);
                    for (uint32_t s = 1; s <= rounds; s++) {
                        wait_seq(mail, s);
                        host_work();
                        for (int k = 0; k < 4; k++) mail->r[k] = (mail->sums[k] ^ (s * 2654435761u)) % bb::P;
                        __atomic_store_n((uint32_t *)&mail->r_seq, s, __ATOMIC_RELEASE);
                    }
                    CHECK(hipStreamSynchronize(st));
                    (fenced ? us_srv : us_srv_ff) = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
                }
                // launches
                mail->seq = 0;
                uint32_t seq = 0;
                t0 = std::chrono::steady_clock::now();
                for (uint32_t layer = 0; layer < layers; layer++) {
                    const Rec *in = d_table;
                    Rec *out = d_a;
                    const EF *eq_l = d_eq;
                    EF r = EF::one();
                    for (uint32_t sr = 0; sr < log_n; sr++) {
                        const uint32_t pairs = (1u << log_n) >> (sr + 1);
                        seq++;
                        hipLaunchKernelGGL(k_round, dim3(1), dim3(threads), 0, st, in, out, eq_l, pairs, sr > 0 ? 1 : 0, r, mail, seq);
                        wait_seq(mail, seq);
                        host_work();
                        for (int k = 0; k < 4; k++) r.c[k].v = (mail->sums[k] ^ (seq * 2654435761u)) % bb::P;
                        if (sr > 0) {
                            in = out;
                            out = out == d_a ? d_b : d_a;
                        }
                        eq_l += pairs;
                    }
                }
                CHECK(hipStreamSynchronize(st));
                us_launch = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
            }
            long long ticks = 0;
            CHECK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
            printf("2^%-2u     %5u     %8.2f (in-kernel %6.2f)   %8.2f          %8.2f              %8.2f\n", log_n, threads, us_dev, ticks / 100.0 / rounds,
                   us_srv, us_srv_ff, us_launch);
        }
    }
    return 0;
}
