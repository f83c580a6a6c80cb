// Throughput of the fp64 VALU ops an FMA-based modular multiply would use (gfx950), same harness
// and units as ubench_int.hip: 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define CHAINS 8
#define DEFD(NAME, BODY)                                                             \
    __global__ void NAME(double *out, double seed) {                                 \
        double x[CHAINS], y = seed, z = 0.5 * seed;                                  \
        for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 1.25 + c + seed;       \
        for (int i = 0; i < ITER; i++) {                                             \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { BODY; }             \
        }                                                                            \
        double s = 0;                                                                \
        for (int c = 0; c < CHAINS; c++) s += x[c];                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + z;                          \
    }
DEFD(k_add, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFD(k_mul, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFD(k_fma, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFD(k_rndne, asm volatile("v_rndne_f64 %0, %0" : "+v"(x[c])))
DEFD(k_cvt, { uint32_t t; asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t) : "v"(x[c])); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x[c]) : "v"(t)); })
DEFD(k_add32, { float t = (float)x[c]; asm volatile("v_add_f32 %0, %0, %1" : "+v"(t) : "v"((float)y)); x[c] = t; })
template <class K> void run(const char *name, K k, int ops_per_body) {
    double *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL(k, grid, block, 0, 0, d, 1.0000001); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k, grid, block, 0, 0, d, 1.0000002); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lane_ops = (double)grid.x * block.x * ITER * CHAINS * ops_per_body;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double wave_instr_per_simd = lane_ops / 64.0 / (p.multiProcessorCount * 4.0);
    double cyc = ms * 1e-3 * p.clockRate * 1e3 / wave_instr_per_simd;
    printf("%-12s %8.3f ms  %8.1f Glane-ops/s  ~%.2f cycles/wave-instr/SIMD (at %d MHz)\n", name, ms, lane_ops / ms / 1e6, cyc, p.clockRate / 1000);
    hipFree(d);
}
int main() {
    run("add_f64", k_add, 1); run("mul_f64", k_mul, 1); run("fma_f64", k_fma, 1); run("rndne_f64", k_rndne, 1);
    run("cvt u32<->f64", k_cvt, 2);
    return 0;
}
