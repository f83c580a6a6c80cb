// Device-side types of the stage kernels (stages.hip). Nothing host-only (no std:: containers) here.
#pragma once
#include "bb31.h"

namespace ovm {
using bb::Fp;
using bb::FpExt;
using EF = bb::FpExt;

// A column of a trace view: value(p) = ptr[(p + rot) & (2^log_height - 1)]  (cyclic lift).
struct ColDesc {
    const Fp *ptr;
    uint32_t log_height;
    uint32_t rot;
};

// Limb-planar arrays (GKR data): limb k of element i is p[k * stride + i]; stride is a multiple
// of 4 and p is 16-byte aligned so 2/4 consecutive elements are one vector access per plane.
struct EFPlanes {
    uint32_t *p;
    uint64_t stride;
};
// The eq weight of pair y in a GKR layer sumcheck, either read from one table (`lo`, already positioned at the launch's
// first pair: lo_bits = EQ_WHOLE) or as the product of two small tables indexed by the halves of the GLOBAL pair index
// y_first + y:  eq(point, y) = eq(point_lo, y mod 2^lo_bits) * eq(point_hi, y >> lo_bits)  (lo_bits = 0: hi alone).
struct EqSplit {
    EFPlanes lo, hi;
    uint32_t lo_bits;
    uint64_t y_first;
};
static constexpr uint32_t EQ_WHOLE = 0xffffffffu;
// Fractions: numerator limbs in 4 planes from p, denominator limbs in 4 planes from p + q_off (normally the
// next four planes; a layer whose numerators are base-field values keeps ONE numerator plane and q_off = stride).
struct FracPlanes {
    uint32_t *p;
    uint64_t stride;
    uint64_t q_off;
    BB_HD static FracPlanes make(uint32_t *p, uint64_t stride) { return FracPlanes{p, stride, 4 * stride}; }
    BB_HD uint32_t *P() const { return p; }
    BB_HD uint32_t *Q() const { return p + q_off; }
    BB_HD FracPlanes at(uint64_t first) const { return FracPlanes{p + first, stride, q_off}; }
};

// Destination of a reduction (stages.hip "last workgroup finishes"): per-workgroup partial
// records, a self-resetting arrival counter, and a host-mapped mailbox (out + flag) that the host
// spins on -- one launch per sumcheck round, no D2H copy, no stream synchronisation.
struct RedOut {
    EF *partials;       // device scratch, >= (number of workgroups) * K
    uint32_t *counter;  // device, zero before the launch; the last workgroup resets it
    EF *out;            // device-visible address of the K results (host-mapped)
    uint32_t *flag;     // device-visible address of the completion flag (host-mapped), may be null
    uint32_t seq;       // value stored to *flag once out[] is visible to the host
};

// ---- stores another agent waits for, without a release FENCE ------------------------------------------------------
// `__threadfence_system()` / `__threadfence()` compile to `buffer_wbl2` + `s_waitcnt` + `buffer_inv`: the wave waits for
// the write-back of EVERY dirty line of its L2 (megabytes right after a fold kernel), then throws the L2's contents
// away. What a mailbox or a partial record needs is only that THESE stores are performed before the flag / counter that
// announces them: store them write-through at the scope of the reader (`sc0 sc1` = system: the host; `sc1` = agent: a
// workgroup on another XCD) and wait for their acknowledgement (`s_waitcnt vmcnt(0)`), which for a write-through store
// arrives from the coherence point of that scope. The READER: the host's loads are ordered by x86; the last workgroup
// reads the records with agent-scope loads (acquire_from_device). OVM_RELEASE_FENCES=1 builds the fenced form.
//
// INVARIANT of the fence-free form: EVERY word a flag or an arrival counter announces is written through st_*_sys /
// st_*_agent (stages.hip: st_ef_sys, st_ef_agent). A plain store added ahead of release_to_host() / release_to_device()
// is NOT ordered by `s_waitcnt vmcnt(0)` against the announcement (it may still sit in this XCD's L2 when the reader
// looks) and nothing diagnoses it -- round 3 found exactly that by parity at 2^20 rows. The fenced build
// (-DOVM_RELEASE_FENCES=1, tools/build_fenced.sh + tests/test_gpu_fenced_build.py) is the cross-check that stays in the suite.
// The wait itself is architecture-specific: on the gfx9 family (gfx90a / gfx942 / gfx950) stores count in vmcnt; from
// gfx10 on they count in vscnt and `s_waitcnt vmcnt(0)` orders nothing -- any other target builds the fenced form.
#ifndef OVM_RELEASE_FENCES
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__))
#define OVM_RELEASE_FENCES 1
#else
#define OVM_RELEASE_FENCES 0
#endif
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !OVM_RELEASE_FENCES && !(defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__))
#error "fence-free mailbox stores rely on gfx9 vmcnt semantics: build this target with -DOVM_RELEASE_FENCES=1"
#endif
__device__ __forceinline__ void st_word_sys(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_word_agent(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stores_performed() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// after the stores to a host mailbox, before its flag
__device__ __forceinline__ void release_to_host() {
#if OVM_RELEASE_FENCES
    __threadfence_system();
#else
    stores_performed();
#endif
}
// the last workgroup, after it has seen the full arrival counter and before it reads the records: nothing here -- the
// records are READ at agent scope (ld_ef_agent: `sc1` loads, which never hit a stale line of this XCD's L2), so the
// `buffer_wbl2` + `buffer_inv` of a fence (write back, then drop, this L2) is not needed either
#ifndef OVM_ACQUIRE_FENCES
#define OVM_ACQUIRE_FENCES OVM_RELEASE_FENCES
#endif
__device__ __forceinline__ void acquire_from_device() {
#if OVM_ACQUIRE_FENCES
    __threadfence();
#endif
}
__device__ __forceinline__ uint32_t ld_word_agent(const uint32_t *p) {
#if OVM_ACQUIRE_FENCES
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// after the stores of a partial record, before the arrival counter
__device__ __forceinline__ void release_to_device() {
#if OVM_RELEASE_FENCES
    __threadfence();
#else
    stores_performed();
#endif
}

static constexpr unsigned MAX_COSETS = 4;
static constexpr unsigned RED_BLOCKS = 2048;  // grid cap of the partial-sum kernels (8 workgroups per CU)
static constexpr unsigned RED_THREADS = 256;

}  // namespace ovm
