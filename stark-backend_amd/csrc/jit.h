// Run-time compiled constraint kernels: per AIR, the round-0 zerocheck / logup evaluation and the
// LogUp-GKR leaf generation as straight-line code (one statement per DAG node the program needs),
// compiled with hiprtc for gfx950 once per proving key. The register-program interpreter
// (program.h, stages.hip) stays as the fallback and as the second implementation the tests compare
// against; both produce identical field elements.
//
// Replaces (reference): the rule encoder + generic evaluator kernels of the CUDA backend
// (crates/cuda-backend/src/logup_zerocheck/rules/, cuda/include/{codec,dag_entry,eval_ctx}.cuh).
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "host.h"

namespace ovm {

struct JitKernels {
    hipFunction_t r0_zerocheck = nullptr;  // null = not compiled (program too large / AIR has none): interpret
    hipFunction_t r0_logup = nullptr;
    hipFunction_t leaves = nullptr;
};

// The loaded code object of one AIR (shared by every key that contains an identical AIR).
struct JitAirModule {
    hipModule_t mod = nullptr;
    JitKernels k;
    ~JitAirModule();
};

struct JitModule {
    std::vector<JitKernels> per_air;
    std::vector<std::shared_ptr<JitAirModule>> modules;  // keep-alive
    double compile_seconds = 0;
};

// Generates and compiles the kernels of every AIR of the key: one translation unit per AIR,
// distinct ones compiled in parallel (OVM_JIT_THREADS, default all cores), compiled code and
// loaded modules cached process-wide by source text. Returns null (and a message) when
// hiprtc is unavailable or the compilation fails; the caller then keeps interpreting.
// load = false: generate and compile only (no device needed), for tests and timing.
std::shared_ptr<JitModule> jit_build(const ProvingKey &pk, std::string &message, bool load = true);

// Number of program instructions jit_build would turn into code (compile time grows with it:
// roughly 5-10 ms each, in one basic block per kernel).
size_t jit_program_size(const ProvingKey &pk);

// Source text of the translation unit jit_build compiles (for inspection and tests).
std::string jit_source(const ProvingKey &pk);

}  // namespace ovm
