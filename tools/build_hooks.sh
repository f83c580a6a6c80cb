#!/usr/bin/env bash
# The TEST TWIN of the library (-DOVM_TEST_HOOKS=1): the product plus the test hooks of csrc/tuning.h -- layout overrides,
# alternative kernels forced on small traces, fault injection of the sharded exchange. The GPU suite loads it through
# OVM_HIP_LIB (tests/conftest.py: pkg_hooks / dev_hooks) NEXT TO the product; the product itself reads none of these names.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "$here/stark-backend_amd/build_hooks"
OVM_OUT="$here/stark-backend_amd/build_hooks/libopenvm_hip_hooks.so" OVM_OBJ="$here/stark-backend_amd/build_hooks/obj" \
  OVM_EXTRA_FLAGS="-DOVM_TEST_HOOKS=1" "$here/stark-backend_amd/build.sh"
