// How should a reduction over a LARGE grid finish? (the 96 large reductions of a config D proof take a
// second launch, sum_partials, + its round trip.) Times, for grids of 256..4096 workgroups of 256 lanes that each
// contribute K extension-field values:
//   none      the kernel alone (each workgroup writes its record with plain stores): what the first launch costs anyway
//   records   + a second launch that sums the records and raises a host flag          (round 2's path for > 64 workgroups)
//   atomics   each workgroup adds its K x 4 Montgomery words into 64-bit accumulators with device-scope atomics and bumps
//             a counter; the LAST workgroup reads the 4K sums, reduces them mod P and raises the flag -- one launch, no
//             pass over the records, no release fence (atomics act at the coherence point)
// and reports launch-to-flag time as the host sees it (spinning on a host-mapped flag), the way the prover waits.
//   hipcc --offload-arch=gfx950 -O3 -I stark-backend_amd/csrc -I include tools/ubench_finish.hip -o /tmp/ubench_finish
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static constexpr uint32_t P = 0x78000001u;

__global__ __launch_bounds__(256) void k_records(uint32_t *records, uint32_t K4, uint32_t spin) {
    uint32_t v = threadIdx.x + blockIdx.x;
    for (uint32_t i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;  // stand-in for the workgroup's real work
    if (threadIdx.x < K4) records[(size_t)blockIdx.x * K4 + threadIdx.x] = v % P;
}
__global__ void k_sum(const uint32_t *records, uint32_t n, uint32_t K4, uint32_t *out, volatile uint32_t *flag, uint32_t seq) {
    const uint32_t k = blockIdx.x;
    unsigned long long acc = 0;
    for (uint32_t b = threadIdx.x; b < n; b += blockDim.x) acc += records[(size_t)b * K4 + k];
    __shared__ unsigned long long sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    __shared__ uint32_t last;
    if (threadIdx.x == 0) {
        out[k] = (uint32_t)(sh[0] % P);
        __threadfence_system();
        last = atomicAdd((uint32_t *)(out + 1024), 1u) == gridDim.x - 1;
        if (last) {
            out[1024] = 0;
            __threadfence_system();
            *flag = seq;
        }
    }
}
template <int COPIES>
__global__ __launch_bounds__(256) void k_atomics(unsigned long long *acc, uint32_t *counter, uint32_t K4, uint32_t spin,
                                                 uint32_t *out, volatile uint32_t *flag, uint32_t seq) {
    uint32_t v = threadIdx.x + blockIdx.x;
    for (uint32_t i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;
    __shared__ uint32_t last;
    unsigned long long seen = 0;
    if (threadIdx.x < K4) {
        // RETURNING atomic (its old value is used below, so the compiler cannot drop the return): the wave's wait for the
        // returned data is the guarantee that the addition has been performed at device scope before the counter moves
        seen = __hip_atomic_fetch_add(acc + (blockIdx.x % COPIES) * 64 + threadIdx.x, (unsigned long long)(v % P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (seen == ~0ull) out[1500] = 1;  // (never true: sums stay below 2^52; keeps the returned value alive)
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x < K4) {
        unsigned long long s = 0;
        for (int c = 0; c < COPIES; c++) s += __hip_atomic_exchange(acc + c * 64 + threadIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out[threadIdx.x] = (uint32_t)(s % P);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = seq;
    }
}

int main() {
    uint32_t *records, *counter, *h_out, *d_out;
    unsigned long long *acc;
    hipMalloc(&records, 4096 * 64 * 4);
    hipMalloc(&counter, 64);
    hipMalloc(&acc, 64 * 64 * 8);
    hipMemset(counter, 0, 64);
    hipMemset(acc, 0, 64 * 64 * 8);
    hipHostMalloc(&h_out, 8192, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void **)&d_out, h_out, 0);
    memset(h_out, 0, 8192);
    volatile uint32_t *h_flag = h_out + 2000;
    uint32_t *d_flag = d_out + 2000;
    uint32_t seq = 0;
    const uint32_t K4 = 12;  // three extension-field values
    printf("{\"K4\": %u, \"rows\": [\n", K4);
    bool first = true;
    for (uint32_t spin : {0u, 2000u})
        for (uint32_t grid : {64u, 256u, 1024u, 2048u, 4096u}) {
            double t[5] = {0, 0, 0, 0, 0};
            uint32_t check[2][12];
            for (int mode = 0; mode < 5; mode++) {
                const int reps = 200;
                double best = 1e9, sum = 0;
                for (int r = 0; r < reps + 20; r++) {
                    hipDeviceSynchronize();
                    auto t0 = std::chrono::steady_clock::now();
                    ++seq;
                    if (mode == 0) {
                        hipLaunchKernelGGL(k_records, dim3(grid), dim3(256), 0, 0, records, K4, spin);
                        hipStreamSynchronize(0);
                    } else if (mode == 1) {
                        hipLaunchKernelGGL(k_records, dim3(grid), dim3(256), 0, 0, records, K4, spin);
                        hipLaunchKernelGGL(k_sum, dim3(K4), dim3(256), 0, 0, records, grid, K4, d_out, d_flag, seq);
                        while (*h_flag != seq) {
                        }
                    } else {
                        if (mode == 2) hipLaunchKernelGGL(k_atomics<1>, dim3(grid), dim3(256), 0, 0, acc, counter, K4, spin, d_out, d_flag, seq);
                        if (mode == 3) hipLaunchKernelGGL(k_atomics<16>, dim3(grid), dim3(256), 0, 0, acc, counter, K4, spin, d_out, d_flag, seq);
                        if (mode == 4) hipLaunchKernelGGL(k_atomics<64>, dim3(grid), dim3(256), 0, 0, acc, counter, K4, spin, d_out, d_flag, seq);
                        while (*h_flag != seq) {
                        }
                    }
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    if (r >= 20) sum += us, best = us < best ? us : best;
                }
                t[mode] = sum / reps;
                if (mode == 1 || mode == 4)
                    for (int k = 0; k < 12; k++) check[mode == 1 ? 0 : 1][k] = h_out[k];
            }
            bool same = true;
            for (int k = 0; k < 12; k++) same &= check[0][k] == check[1][k];
            printf("%s  {\"spin\": %u, \"workgroups\": %u, \"kernel_and_sync_us\": %.2f, \"records_plus_second_launch_us\": %.2f, "
                   "\"atomics_one_launch_us\": %.2f, \"atomics_16_copies_us\": %.2f, \"atomics_64_copies_us\": %.2f, \"same_sums\": %s}",
                   first ? "" : ",\n", spin, grid, t[0], t[1], t[2], t[3], t[4], same ? "true" : "false");
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
