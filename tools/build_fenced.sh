#!/usr/bin/env bash
# The FENCED twin of the library (device_common.h: -DOVM_RELEASE_FENCES=1 puts __threadfence / __threadfence_system back
# in front of every mailbox flag and arrival counter, and the acquire fences with them): the cross-check of the
# fence-free default, loaded by tests/test_gpu_fenced_build.py through OVM_HIP_LIB. Built next to the product, never
# loaded by it.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "$here/stark-backend_amd/build_fenced"
OVM_OUT="$here/stark-backend_amd/build_fenced/libopenvm_hip_fenced.so" OVM_OBJ="$here/stark-backend_amd/build_fenced/obj" \
  OVM_EXTRA_FLAGS="-DOVM_RELEASE_FENCES=1" "$here/stark-backend_amd/build.sh"
