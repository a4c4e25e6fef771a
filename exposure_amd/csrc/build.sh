#!/bin/bash
# Build libexposure_hip.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun).
# Two translation units: the streaming kernels (default flags) and the VALU-bound fused inference
# kernel (-fno-slp-vectorize -fno-honor-nans, see chain_fused.hip); extra arguments go to both compile steps.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${EXPO_LIB_OUT:-$HERE/../libexposure_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC)
"$HIPCC" "${FLAGS[@]}" "$@" -c "$HERE/exposure_hip.hip" -o "$TMP/exposure_hip.o" &
"$HIPCC" "${FLAGS[@]}" -fno-slp-vectorize -fno-honor-nans "$@" -c "$HERE/chain_fused.hip" -o "$TMP/chain_fused.o" &
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$TMP/exposure_hip.o" "$TMP/chain_fused.o" -o "$OUT"
echo "built $OUT"
