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
    // 64-bit accumulators (device, zero before the launch, reset by the last workgroup), or null. Non-null = the
    // "atomic finish" of a LARGE grid: every workgroup adds the Montgomery words of its K totals into acc[4K] with
    // device-scope atomics instead of storing a record; the last workgroup reduces the sums mod P into out[].
    unsigned long long *acc;
};
static constexpr unsigned RED_ATOMIC_MAX_K = 16;  // acc[] holds 4 * 16 words

static constexpr unsigned MAX_COSETS = 4;
static constexpr unsigned RED_BLOCKS = 2048;  // grid cap of the partial-sum kernels (8 workgroups per CU)
static constexpr unsigned RED_THREADS = 256;

}  // namespace ovm
