// HBM streaming microbenchmark for gfx950: read-only sum, copy and write-only fill with 16-byte
// accesses per lane, at several grid sizes. Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mem.hip -o ubench_mem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void rd(const uint4 *in, uint64_t n, uint32_t *out) {
    uint64_t stride = (uint64_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) out[0] = acc;
}
// 4 independent loads in flight per lane
__global__ __launch_bounds__(256) void rd4(const uint4 *in, uint64_t n, uint32_t *out) {
    uint64_t stride = (uint64_t)gridDim.x * 256;
    uint32_t acc = 0;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n; i += stride) { uint4 v = in[i]; acc += v.x ^ v.w; }
    if (acc == 0x12345678) out[0] = acc;
}
__global__ __launch_bounds__(256) void cp(const uint4 *in, uint4 *o, uint64_t n) {
    uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) o[i] = in[i];
}
__global__ __launch_bounds__(256) void wr(uint4 *o, uint64_t n) {
    uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) o[i] = make_uint4(i, 1, 2, 3);
}
// one element per lane, no loop (grid covers the array)
__global__ __launch_bounds__(256) void cp1(const uint4 *in, uint4 *o, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = in[i];
}

int main() {
    const uint64_t bytes = 4ull << 30, n = bytes / 16;
    uint4 *a, *b; uint32_t *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto f) { f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 5; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5; };
    for (unsigned g : {1024u, 2048u, 4096u, 8192u, 16384u, 65536u}) {
        float t1 = time([&] { hipLaunchKernelGGL(rd, dim3(g), dim3(256), 0, 0, a, n, o); });
        float t4 = time([&] { hipLaunchKernelGGL(rd4, dim3(g), dim3(256), 0, 0, a, n, o); });
        float t2 = time([&] { hipLaunchKernelGGL(cp, dim3(g), dim3(256), 0, 0, a, b, n); });
        float t3 = time([&] { hipLaunchKernelGGL(wr, dim3(g), dim3(256), 0, 0, b, n); });
        printf("grid %6u: read %.0f GB/s  read4 %.0f GB/s  copy %.0f GB/s (r+w)  write %.0f GB/s\n", g, bytes / t1 / 1e6, bytes / t4 / 1e6, 2.0 * bytes / t2 / 1e6, bytes / t3 / 1e6);
    }
    float t5 = time([&] { hipLaunchKernelGGL(cp1, dim3(n / 256), dim3(256), 0, 0, a, b, n); });
    printf("copy, one 16B element per lane, grid %llu: %.0f GB/s (r+w)\n", (unsigned long long)(n / 256), 2.0 * bytes / t5 / 1e6);
    float t6 = time([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpy D2D: %.0f GB/s (r+w)\n", 2.0 * bytes / t6 / 1e6);
    return 0;
}
