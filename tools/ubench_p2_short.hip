// Does the permutation's throughput depend on how long a wave lives? The Merkle kernels run 1 (compress) or 4 (row hash of
// 25 columns) permutations per lane and then the wave ends; the ceiling harness (ubench_poseidon2.hip) runs 256. Same
// total work (2^27 permutations), `iters` permutations per lane, grid sized to match; states come from memory and the
// digest half goes back (32 B in, 32 B out per lane and iteration group), like the kernels.
//   hipcc --offload-arch=gfx950 -O3 -I stark-backend_amd/csrc -I include tools/ubench_p2_short.hip -o tools/ubench_p2_short
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "poseidon2.h"
using bb::Fp;
__global__ __launch_bounds__(256) void k_short(const uint4 *__restrict__ in, uint4 *__restrict__ out, int iters) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp s[16];
    const uint4 a = in[2 * t], b = in[2 * t + 1];
    s[0] = Fp::raw(a.x), s[1] = Fp::raw(a.y), s[2] = Fp::raw(a.z), s[3] = Fp::raw(a.w);
    s[4] = Fp::raw(b.x), s[5] = Fp::raw(b.y), s[6] = Fp::raw(b.z), s[7] = Fp::raw(b.w);
#pragma unroll
    for (int i = 8; i < 16; i++) s[i] = Fp::zero();
#pragma nounroll
    for (int it = 0; it < iters; it++) p2::permute(s);
    out[2 * t] = make_uint4(s[0].v, s[1].v, s[2].v, s[3].v);
    out[2 * t + 1] = make_uint4(s[4].v, s[5].v, s[6].v, s[7].v);
}
// persistent form: the grid fills the chip once (blocks = CUs * 8), each lane walks `per_lane` states `stride` apart
__global__ __launch_bounds__(256) void k_persistent(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, int iters) {
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        Fp s[16];
        const uint4 a = in[2 * t], b = in[2 * t + 1];
        s[0] = Fp::raw(a.x), s[1] = Fp::raw(a.y), s[2] = Fp::raw(a.z), s[3] = Fp::raw(a.w);
        s[4] = Fp::raw(b.x), s[5] = Fp::raw(b.y), s[6] = Fp::raw(b.z), s[7] = Fp::raw(b.w);
#pragma unroll
        for (int i = 8; i < 16; i++) s[i] = Fp::zero();
#pragma nounroll
        for (int it = 0; it < iters; it++) p2::permute(s);
        out[2 * t] = make_uint4(s[0].v, s[1].v, s[2].v, s[3].v);
        out[2 * t + 1] = make_uint4(s[4].v, s[5].v, s[6].v, s[7].v);
    }
}
int main() {
    const size_t total = (size_t)1 << 27;
    uint4 *in, *out;
    (void)hipMalloc(&in, total * 32);
    (void)hipMalloc(&out, total * 32);
    (void)hipMemset(in, 1, total * 32);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    for (int iters : {1, 2, 4, 16, 64, 256}) {
        const size_t lanes = total / iters;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k_short, dim3(lanes / 256), dim3(256), 0, 0, in, out, iters);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("one wave = %3d permutations per lane, %8zu workgroups: %7.3f ms -> %6.3f Gperm/s\n", iters, lanes / 256, best, total / (best * 1e-3) / 1e9);
    }
    for (int iters : {1, 4}) {
        for (int mult : {4, 8, 16}) {
            const size_t lanes = total / iters;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(k_persistent, dim3(p.multiProcessorCount * mult), dim3(256), 0, 0, in, out, lanes, iters);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms;
                (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("persistent, %d permutations per state, %d workgroups per CU: %7.3f ms -> %6.3f Gperm/s\n", iters, mult, best, total / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
