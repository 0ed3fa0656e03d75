#!/bin/bash
# A/B builds for kernel experiments: tools/ab_build.sh TAG [extra hipcc flags, e.g. -DAMT_EXPERIMENT to make the library read the AMT_* tuning variables]  ->  video2music_amd/lib/libamt_hip.TAG.so
# (select at run time with AMT_LIB=video2music_amd/lib/libamt_hip.TAG.so; the default library is untouched)
set -euo pipefail
TAG="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/video2music_amd/csrc"
OBJ="$SRC/obj_$TAG"
mkdir -p "$OBJ" "$ROOT/video2music_amd/lib"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $*"
pids=(); objs=()
for s in "$SRC"/*.hip; do
  o="$OBJ/$(basename "${s%.hip}").o"; objs+=("$o")
  $HIPCC $FLAGS -c "$s" -o "$o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$ROOT/video2music_amd/lib/libamt_hip.$TAG.so" "${objs[@]}"
echo "built libamt_hip.$TAG.so"
