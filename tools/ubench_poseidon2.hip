// Measured ceiling of the Poseidon2 instruction mix on gfx950: the SAME permutation code the Merkle
// kernels run (csrc/poseidon2.h, p2::permute) on register-resident states -- ITER chained permutations
// per lane, no memory traffic inside the loop -- at the occupancy of hash_rows_kernel (256-lane
// workgroups, as many waves per SIMD as the register budget gives). What hash_rows_kernel<1> achieves
// (bench.py: roofline.issue_bound.achieved_Gperm_s) is quoted against this number: the difference is
// what loads, stores and the sponge bookkeeping cost; anything beyond it needs FEWER INSTRUCTIONS per
// permutation, not better scheduling. Also reads s_memtime around the loop to report the shader clock
// the run actually had and the cycles one wave-level VALU instruction took.
//   hipcc --offload-arch=gfx950 -O3 -I stark-backend_amd/csrc -I include tools/ubench_poseidon2.hip -o tools/ubench_poseidon2
//   tools/ubench_poseidon2 [json path]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "poseidon2.h"

using bb::Fp;
static constexpr int ITER = 256;
static constexpr int VALU_PER_PERM = 3676;  // the full permutation (p2::permute = permute_lazy_t on the device), static count (tools/isa_count.py);
                                             // rounds 3-5: 7218 (one reduction per addition, five-instruction products)

// ticks[0] = shader cycles (s_memtime), ticks[1] = 100 MHz reference ticks (s_memrealtime) of one wave's loop, taken in a
// workgroup from the middle of the grid (every SIMD is full then): the clock the loop ran at is their ratio x 100 MHz
// (MI355X_MICROARCH.md, DVFS item 6).
__global__ __launch_bounds__(256) void perm_loop_kernel(uint32_t *out, uint32_t seed, unsigned long long *ticks) {
    Fp s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = Fp::raw((threadIdx.x * 2654435761u + blockIdx.x * 40503u + i * 97u + seed) % bb::P);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; it++) p2::permute(s);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) x ^= s[i].v;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {
        ticks[0] = t1 - t0;
        ticks[1] = r1 - r0;
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8 * 8;  // several rounds of (up to) 8 resident workgroups per CU
    uint32_t *d;
    unsigned long long *d_ticks, ticks[2] = {0, 0};
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipMalloc(&d_ticks, 16);
    int blocks_per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, perm_loop_kernel, 256, 0);
    const double waves_per_simd = blocks_per_cu;  // a 256-lane workgroup = 4 waves, one per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(perm_loop_kernel, dim3(blocks), dim3(256), 0, 0, d, 1u, d_ticks);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(perm_loop_kernel, dim3(blocks), dim3(256), 0, 0, d, 3u + rep, d_ticks);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipMemcpy(ticks, d_ticks, 16, hipMemcpyDeviceToHost);
    const double wave_instr_per_simd = (double)blocks * 256 * ITER * VALU_PER_PERM / 64.0 / (p.multiProcessorCount * 4.0);
    const double clock_ghz = ticks[1] ? (double)ticks[0] / (double)ticks[1] * 0.1 : 0.0;
    // shader cycles per wave-level VALU instruction per SIMD over the whole launch, at the clock the loop really had
    const double real_cyc = best * 1e-3 * clock_ghz * 1e9 / wave_instr_per_simd;
    // how many waves shared the stamped wave's SIMD: its loop took this many instruction slots
    // (the occupancy API's answer, `waves_per_simd`, is an upper bound: MI355X_MICROARCH.md, correctness boundaries)
    const double co_resident = real_cyc > 0 ? ticks[0] / (ITER * (double)VALU_PER_PERM * real_cyc) : 0.0;
    const double perms = (double)blocks * 256 * ITER;
    const double gperm = perms / (best * 1e-3) / 1e9;
    const double cyc_nominal = best * 1e-3 * 2.4e9 / wave_instr_per_simd;
    // s_memtime counts at a fixed 100 MHz on gfx9: wall time of one wave's loop, for reference
    printf("poseidon2 mix ceiling: %.3f ms for %.3g permutations -> %.3f Gperm/s; %.2f cycles per wave-level VALU instruction per SIMD at a nominal "
           "2.4 GHz (%d CUs, %d blocks of 256 lanes, %d permutations per lane)\n"
           "in-kernel clock (s_memtime / s_memrealtime): %.3f GHz -> %.2f shader cycles per wave-level VALU instruction per SIMD; "
           "the stamped wave shared its SIMD with %.1f waves in all (occupancy API: up to %.0f)\n",
           best, perms, gperm, cyc_nominal, p.multiProcessorCount, blocks, ITER, clock_ghz, real_cyc, co_resident, waves_per_simd);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        if (f) {
            fprintf(f, "{\"gperm_per_s\": %.4f, \"ms\": %.4f, \"permutations\": %.0f, \"valu_instr_per_permutation\": %d, "
                       "\"cycles_per_wave_valu_instr_per_simd_at_2.4GHz\": %.3f, \"in_kernel_clock_GHz\": %.3f, \"resident_waves_per_simd\": %.0f, "
                       "\"shader_cycles_per_wave_valu_instr_per_simd\": %.3f, \"compute_units\": %d, \"source\": \"tools/ubench_poseidon2.hip\"}\n",
                    gperm, best, perms, VALU_PER_PERM, cyc_nominal, clock_ghz, co_resident, real_cyc, p.multiProcessorCount);
            fclose(f);
        }
    }
    hipFree(d);
    hipFree(d_ticks);
    return 0;
}
