// Issue cost of the 64-bit / three-operand integer VALU ops a lazily reduced Poseidon2 linear layer could be
// built from (gfx950): v_lshl_add_u64, the v_add_co / v_addc pair, v_add3_u32, v_mad_i64_i32, v_lshlrev_b64,
// v_ashrrev_i32, v_alignbit_b32, next to v_add_u32 / v_mul_hi_i32 as the yardsticks. Same harness as
// tools/ubench_int.hip: ITER iterations of 8 independent chains per lane, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define CHAINS 8
#define DEF32(NAME, ...)                                                            \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                             \
        uint32_t x[CHAINS], y = seed | 1u, z = seed * 7u + 3u;                       \
        for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 2654435761u + c + seed;\
        for (int i = 0; i < ITER; i++) {                                             \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { __VA_ARGS__; }             \
        }                                                                            \
        uint32_t s = z;                                                              \
        for (int c = 0; c < CHAINS; c++) s ^= x[c];                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                              \
    }
#define DEF64(NAME, ...)                                                            \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                             \
        uint64_t x[CHAINS], y = ((uint64_t)seed << 32) | (seed | 1u);                \
        uint32_t w = seed | 1u;                                                      \
        for (int c = 0; c < CHAINS; c++) x[c] = (uint64_t)(threadIdx.x * 2654435761u + c + seed) * 0x9E3779B97F4A7C15ull;\
        for (int i = 0; i < ITER; i++) {                                             \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { __VA_ARGS__; }             \
        }                                                                            \
        uint64_t s = w;                                                              \
        for (int c = 0; c < CHAINS; c++) s ^= x[c];                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32); \
    }
DEF32(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF32(k_sub, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF32(k_min, asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF32(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEF32(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF32(k_mul_hi_i, asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEF32(k_ashr, asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(x[c])))
DEF32(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %1, 5" : "+v"(x[c]) : "v"(y)))
DEF32(k_med3, asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEF32(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEF32(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEF64(k_lshl_add64_0, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[c]) : "v"(y)))
DEF64(k_lshl_add64_1, asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(x[c]) : "v"(y)))
DEF64(k_addc_pair, { uint32_t lo = (uint32_t)x[c], hi = (uint32_t)(x[c] >> 32);
    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"((uint32_t)y), "v"((uint32_t)(y >> 32)) : "vcc");
    x[c] = ((uint64_t)hi << 32) | lo; })
DEF64(k_mad_i64, asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"((uint32_t)x[c]), "v"(w) : "vcc"))
DEF64(k_mad_u64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"((uint32_t)x[c]), "v"(w) : "vcc"))
DEF64(k_lshl64, asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x[c])))
DEF64(k_ashr64, asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(x[c])))
DEF64(k_pk_mov, asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(x[c]) : "v"(y)))
DEF64(k_pk_add_f32, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
// a mixed stream the way a lazily reduced layer would issue it: 4 64-bit adds, then a signed Montgomery reduction (3)
DEF64(k_mix, {
    asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %0, %0, 1, %1\n\tv_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %0, %0, 0, %1"
                 : "+v"(x[c]) : "v"(y));
    uint32_t lo = (uint32_t)x[c], hi = (uint32_t)(x[c] >> 32);
    asm volatile("v_mul_lo_u32 %0, %0, %2\n\tv_mul_hi_i32 %0, %0, %2\n\tv_sub_u32 %0, %1, %0" : "+v"(lo) : "v"(hi), "v"(w));
    x[c] = ((uint64_t)hi << 32) | lo; })
template <class K> void run(const char *name, K k, int ops_per_body) {
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL(k, grid, block, 0, 0, d, 1u); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k, grid, block, 0, 0, d, 3u); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    double lane_ops = (double)grid.x * block.x * ITER * CHAINS * ops_per_body;
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    double wave_instr_per_simd = lane_ops / 64.0 / (p.multiProcessorCount * 4.0);
    double cyc = best * 1e-3 * p.clockRate * 1e3 / wave_instr_per_simd;
    printf("%-16s %8.3f ms  %8.1f Glane-ops/s  ~%.2f cycles/wave-instr/SIMD (at %d MHz)\n", name, best, lane_ops / best / 1e6, cyc, p.clockRate / 1000);
    (void)hipFree(d);
}
int main() {
    run("add_u32", k_add, 1); run("sub_u32", k_sub, 1); run("min_u32", k_min, 1); run("add3_u32", k_add3, 1);
    run("mul_lo_u32", k_mul_lo, 1); run("mul_hi_i32", k_mul_hi_i, 1); run("ashrrev_i32", k_ashr, 1);
    run("alignbit_b32", k_alignbit, 1); run("med3_i32", k_med3, 1); run("and_or_b32", k_and_or, 1); run("perm_b32", k_perm, 1);
    run("lshl_add_u64 <<0", k_lshl_add64_0, 1); run("lshl_add_u64 <<1", k_lshl_add64_1, 1);
    run("add_co+addc", k_addc_pair, 2); run("mad_i64_i32", k_mad_i64, 1); run("mad_u64_u32", k_mad_u64, 1);
    run("lshlrev_b64", k_lshl64, 1); run("ashrrev_i64", k_ashr64, 1); run("pk_mov_b32", k_pk_mov, 1); run("pk_add_f32", k_pk_add_f32, 1);
    run("mix 4add64+redc", k_mix, 7);
    return 0;
}
