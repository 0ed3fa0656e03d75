#!/bin/bash
# Builds libamt_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $*"
objs=()
pids=()
for src in "$HERE"/*.hip; do
  obj="$HERE/obj/$(basename "${src%.hip}").o"
  objs+=("$obj")
  stale=0
  for hdr in "$HERE"/*.h "$HERE/../../include/amt_hip.h"; do [ "$hdr" -nt "$obj" ] && stale=1; done
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ $stale = 1 ]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libamt_hip.so" "${objs[@]}"
echo "built $OUT/libamt_hip.so"
