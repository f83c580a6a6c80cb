#!/usr/bin/env bash
# ONE COMMAND for a machine that has cargo (and, for step 3, an MI355X with ROCm): pins the oracle to the reference and
# runs the reference's own conformance suite on the HIP engine.
#
#   tools/run_on_cargo_box.sh /path/to/checkout-of-openvm-org-stark-backend-v2.0.1 [--no-gpu]
#
# 1. golden vectors:   copies tools/gen_golden.rs into the checkout as an example of openvm-stark-sdk and runs it with
#                      RAYON_NUM_THREADS=1 (sequential find_any = the MINIMUM proof-of-work witness, which is what this
#                      backend returns) -> tests/golden/reference/*.json: one document per case of tests/conformance.py
#                      (key blob, traces, Proof::encode_to_vec of BabyBearPoseidon2CpuEngine) + the default Poseidon2 KAT.
# 2. pins:             python -m pytest tests/test_oracle_golden.py  (CPU): the oracle verifies and REPRODUCES every
#                      reference proof byte for byte; the restated fixtures equal the reference's. With a GPU also
#                      tests/test_gpu_prove.py::test_reference_golden_proofs (the HIP prover reproduces them).
# 3. conformance:      drops crates/hip-backend into the checkout's workspace, builds it against libopenvm_hip.so and runs
#                      `backend_test_suite!(HipEngine)` + the bit-identity tests (crates/hip-backend/tests/integration.rs).
# Nothing here has been run by the builder of this repository (no cargo in that image): expect to fix import paths in
# gen_golden.rs / the crate on first contact and say so in the commit that adds tests/golden/reference/.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF="${1:?usage: tools/run_on_cargo_box.sh <reference checkout> [--no-gpu]}"
GPU=1; [ "${2:-}" = "--no-gpu" ] && GPU=0
command -v cargo >/dev/null || { echo "cargo not found"; exit 2; }
[ -f "$REF/crates/stark-sdk/Cargo.toml" ] || { echo "$REF is not a checkout of openvm-org/stark-backend"; exit 2; }
( cd "$REF" && git describe --tags --always 2>/dev/null | sed 's/^/reference at: /' ) || true

echo "== 1. golden vectors"
mkdir -p "$REF/crates/stark-sdk/examples"
cp "$HERE/tools/gen_golden.rs" "$REF/crates/stark-sdk/examples/gen_golden.rs"
OUT="$HERE/tests/golden/reference"
mkdir -p "$OUT"
( cd "$REF" && RAYON_NUM_THREADS=1 RUSTFLAGS="-Ctarget-cpu=native" \
    cargo run --release -p openvm-stark-sdk --features openvm-stark-backend/test-utils --example gen_golden -- "$OUT" )
ls "$OUT" | wc -l | sed 's/^/documents written: /'

echo "== 2. the oracle against the reference's bytes"
( cd "$HERE" && make -s -C oracle && python -m pytest tests/test_oracle_golden.py -q )

if [ "$GPU" = 1 ]; then
  echo "== 2b. the HIP prover against the reference's bytes"
  ( cd "$HERE" && ./stark-backend_amd/build.sh && python -m pytest tests/test_gpu_prove.py -q -m gpu -k reference_golden )
  echo "== 3. backend_test_suite!(HipEngine)"
  rm -rf "$REF/crates/hip-backend"
  cp -r "$HERE/crates/hip-backend" "$REF/crates/hip-backend"
  grep -q '"crates/hip-backend"' "$REF/Cargo.toml" || \
    sed -i 's|    "crates/cuda-backend",|    "crates/cuda-backend",\n    "crates/hip-backend",|' "$REF/Cargo.toml"
  grep -q '^openvm-hip-backend' "$REF/Cargo.toml" || \
    sed -i 's|^openvm-cuda-backend = .*|&\nopenvm-hip-backend = { path = "crates/hip-backend", version = "2.0.1", default-features = false }|' "$REF/Cargo.toml"
  ( cd "$REF" && OPENVM_HIP_LIB_DIR="$HERE/stark-backend_amd" RAYON_NUM_THREADS=1 \
      cargo test --release -p openvm-hip-backend -- --test-threads=1 )
fi
echo "done: commit tests/golden/reference/ (the documents are data: inputs and the reference's outputs)"
