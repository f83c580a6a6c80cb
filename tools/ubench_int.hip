// Throughput of the integer VALU ops a BabyBear Montgomery multiply can be built from (gfx950).
// Each kernel runs ITER iterations of 8 independent chains per lane; prints Gops/s chip-wide and
// the implied issue cycles per wave-instruction per SIMD (4 SIMDs/CU, 256 CUs, clock from props).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define CHAINS 8
#define DEF(NAME, BODY)                                                              \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                             \
        uint32_t x[CHAINS], y = seed | 1u;                                           \
        for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 2654435761u + c + seed;\
        for (int i = 0; i < ITER; i++) {                                             \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { BODY; }             \
        }                                                                            \
        uint32_t s = 0;                                                              \
        for (int c = 0; c < CHAINS; c++) s ^= x[c];                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                              \
    }
DEF(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 27, %1" : "+v"(x[c]) : "v"(y)))
DEF(k_min, asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF(k_mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[c]) : "v"(y)))
DEF(k_mad_u64, { uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x[c]), "v"(y) : "vcc"); x[c] = (uint32_t)(r >> 32) ^ (uint32_t)r; })
template <class K> void run(const char *name, K k, int ops_per_body) {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL(k, grid, block, 0, 0, d, 1u); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k, grid, block, 0, 0, d, 3u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lane_ops = (double)grid.x * block.x * ITER * CHAINS * ops_per_body;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double wave_instr_per_simd = lane_ops / 64.0 / (p.multiProcessorCount * 4.0);
    double cyc = ms * 1e-3 * p.clockRate * 1e3 / wave_instr_per_simd;
    printf("%-12s %8.3f ms  %8.1f Glane-ops/s  ~%.2f cycles/wave-instr/SIMD (at %d MHz)\n", name, ms, lane_ops / ms / 1e6, cyc, p.clockRate / 1000);
    hipFree(d);
}
int main() {
    run("add_u32", k_add, 1); run("lshl_add", k_lshl_add, 1); run("min_u32", k_min, 1);
    run("mul_lo_u32", k_mul_lo, 1); run("mul_hi_u32", k_mul_hi, 1); run("mul_u32_u24", k_mul_u24, 1);
    run("mad_u32_u24", k_mad_u24, 1); run("mad_u64_u32", k_mad_u64, 1);
    return 0;
}
