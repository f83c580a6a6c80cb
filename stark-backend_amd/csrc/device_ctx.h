// Per-process device context: twiddle tables and small helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>

#include "bb31.h"

#define OVM_CHECK(expr)                                   \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) return (int)_e;             \
    } while (0)

// Kernel launches report through hipGetLastError like the reference's CHECK_KERNEL()
// (crates/cuda-common/include/launcher.cuh:41-54).
#define OVM_CHECK_LAUNCH() OVM_CHECK(hipGetLastError())

// Every kernel launch of the library goes through OVM_LAUNCH: small host-to-device uploads are not copied one by
// one but collected (prover.hip: Ctx::upload) and written by ONE scatter kernel right before the next launch on
// their stream -- a 36-AIR proof made ~700 separate copies of a few hundred bytes each, ~15 us of stream time apiece.
namespace ovm {
void flush_uploads(hipStream_t stream);
// Per-thread device resources (result mailboxes, the pinned upload arena, the grind result word, the memory pools)
// are released by the thread's exit -- or at once by ovm_hip_thread_shutdown(). Their destructors must not call into
// a HIP runtime that is already being torn down: runtime_exiting() turns true in an atexit handler registered on
// first use of the library (i.e. after the runtime's own, so it runs before them), and the destructors then leave the
// memory to the dying process.
bool runtime_exiting();
struct ThreadResource {  // a TLS object with device memory: release() frees it now (idempotent)
    ThreadResource();
    virtual ~ThreadResource();
    virtual void release() = 0;
};
}
#define OVM_LAUNCH(kernel, grid, block, shmem, stream, ...)                        \
    do {                                                                           \
        ovm::flush_uploads(stream);                                                \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);       \
    } while (0)

namespace ovm {

static constexpr unsigned TW_LOG_MAX = 27;  // two-adicity of BabyBear
static constexpr unsigned TW_LO_BITS = 12;
static constexpr unsigned TW_HI_BITS = TW_LOG_MAX - TW_LO_BITS;

// omega = two_adic_generator(27).  lo[e] = omega^e (e < 2^12),  hi[e] = omega^(e << 12).
// Any root of unity of order 2^k | 2^27 is  omega^(e << (27-k)) = hi[..] * lo[..]: two L2-resident
// loads and one multiply instead of a 2^k-entry table (a 2^25-point transform would need 128 MB).
struct Twiddles {
    const bb::Fp *lo;
    const bb::Fp *hi;
};

__device__ __forceinline__ bb::Fp tw_pow27(const Twiddles &tw, uint32_t e27) {
    e27 &= (1u << TW_LOG_MAX) - 1;
    bb::Fp l = tw.lo[e27 & ((1u << TW_LO_BITS) - 1)];
    bb::Fp h = tw.hi[e27 >> TW_LO_BITS];
    return l * h;
}

// Returns the table for the current device (building it on first use). nullptr on failure.
const Twiddles *twiddles();

// Optional per-stage timing with HIP events recorded on the launching stream (bench.py reads it
// through ovm_hip_profile_*). Off by default: no events are created.
// the library's side stream of the calling thread (prover.hip: SideStream): scopes recorded on it are reported apart
void prof_register_side_stream(hipStream_t s);
struct ProfScope {
    int slot = -1;
    unsigned generation = 0;  // of the record list when the scope opened: profile_enable clears the list and bumps it
    bool ranged = false;  // a roctx range is open (OVM_ROCTX=1)
    hipStream_t stream;
    ProfScope(const char *label, hipStream_t s);
    ~ProfScope();
};
#define OVM_PROF_CAT2(a, b) a##b
#define OVM_PROF_CAT(a, b) OVM_PROF_CAT2(a, b)
#define OVM_PROF(label, stream) ovm::ProfScope OVM_PROF_CAT(_prof_scope_, __LINE__)(label, stream)

inline unsigned ceil_div(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

__host__ __device__ __forceinline__ uint32_t bitrev32(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

}  // namespace ovm
