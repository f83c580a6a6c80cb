#!/usr/bin/env bash
# Builds libopenvm_hip.so (gfx950 only) in-tree. hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
src="$here/csrc"
out="${OVM_OUT:-$here/libopenvm_hip.so}"  # (OVM_OUT / OVM_OBJ / OVM_EXTRA_FLAGS: A/B builds of a variant next to the product)
obj="${OVM_OBJ:-$here/build}"
mkdir -p "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DOVM_HOST_FAST_PERMUTE=1 -I"$here/../include" -I"$src" -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-result ${OVM_EXTRA_FLAGS:-})
pids=()
objs=()
for f in "$src"/*.hip "$src"/*.cpp; do
  [ -e "$f" ] || continue
  o="$obj/$(basename "$f").o"
  objs+=("$o")
  if [ ! -e "$o" ] || [ -n "$(find "$src" "$here/../include" -newer "$o" \( -name '*.h' -o -name '*.hpp' -o -name '*.cuh' -o -name '*.inc' \) -print -quit)" ] || [ "$f" -nt "$o" ]; then
    "$HIPCC" "${FLAGS[@]}" -x hip -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
