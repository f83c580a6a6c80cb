// 16-row chunks (64 bytes) of a column, one per lane, moved between global memory and the lanes through LDS.
// Device code only.
//
// A lane that owns a chunk reads it as four 16-byte loads, and a wave's four load instructions then each touch 64
// SEPARATE 64-byte segments, 16 bytes of each: the memory pipeline handles a quarter of the bytes per request it could.
// The same bytes as four instructions in which lane l moves bytes [16 l, 16 l + 16) of a 1 KiB run are whole-line
// requests; the pieces change lanes in LDS (a chunk's four pieces 80 bytes apart, so that neither the run-order nor
// the chunk-order accesses collide on banks). Measured on the coset LDE (config B, 2 859 columns x 2^20 rows):
// 11.0 -> 7.1 ms, 3.2 -> 4.9 TB/s. The wave's 64 chunks must be contiguous in memory (4 KiB); LDS operations of one
// wave execute in order, so no workgroup barrier is involved.
#pragma once
#include <hip/hip_runtime.h>

namespace ovm {

static constexpr int WAVE_CHUNK_TILE = 64 * 5;  // uint4 per wave

// T: this wave's tile; src_wave: the first of the wave's 64 chunks. own[v] = bytes [16 v, 16 v + 16) of lane's chunk.
__device__ __forceinline__ void wave_chunks_load(uint4 *T, const uint4 *src_wave, uint32_t lane, uint4 (&own)[4]) {
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const uint32_t p = v * 64 + lane;  // piece p of the run: chunk p / 4, part p % 4
        T[(p >> 2) * 5 + (p & 3)] = src_wave[p];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int v = 0; v < 4; v++) own[v] = T[lane * 5 + v];
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_chunks_store(uint4 *T, uint4 *dst_wave, uint32_t lane, const uint4 (&own)[4]) {
#pragma unroll
    for (int v = 0; v < 4; v++) T[lane * 5 + v] = own[v];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const uint32_t p = v * 64 + lane;
        dst_wave[p] = T[(p >> 2) * 5 + (p & 3)];
    }
    __builtin_amdgcn_wave_barrier();
}

}  // namespace ovm
