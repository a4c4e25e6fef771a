#!/bin/bash
# Build libexposure_hip.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun).
# Seven translation units (critic_step.hip: the reductions of the hand-scheduled critic update; conv_ops.hip: the convnets' 4x4 / stride-2 convolution on the f32 matrix cores; curve_generic.hip: Tone / Color for cfg.curve_steps other than 8): the streaming kernels (default flags), the VALU-bound fused inference kernel
# (-fno-slp-vectorize -fno-honor-nans, see chain_fused.hip), the convnets' activation (nn_ops.hip) and the one-pass
# backward of a fixed sequence (chain_fused_bwd.hip; -fno-slp-vectorize: the packed-fp32 pairs cost it ~100 VGPRs); extra
# arguments go to every compile step.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${EXPO_LIB_OUT:-$HERE/../libexposure_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
# the digest of the sources, baked into the binary (expo_build_info): _cabi.load() recomputes it and refuses a stale .so
DIGEST="$(cd "$HERE" && LC_ALL=C ls *.hip *.h build.sh | LC_ALL=C sort | xargs cat ../../include/exposure_hip.h | sha256sum | cut -d' ' -f1)"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC "-DEXPO_SOURCE_DIGEST=\"$DIGEST\"")
"$HIPCC" "${FLAGS[@]}" "$@" -c "$HERE/exposure_hip.hip" -o "$TMP/exposure_hip.o" &
p1=$!
"$HIPCC" "${FLAGS[@]}" -fno-slp-vectorize -fno-honor-nans "$@" -c "$HERE/chain_fused.hip" -o "$TMP/chain_fused.o" &
p2=$!
"$HIPCC" "${FLAGS[@]}" "$@" -c "$HERE/nn_ops.hip" -o "$TMP/nn_ops.o" &
p3=$!
"$HIPCC" "${FLAGS[@]}" -fno-slp-vectorize "$@" -c "$HERE/chain_fused_bwd.hip" -o "$TMP/chain_fused_bwd.o" &
p4=$!
"$HIPCC" "${FLAGS[@]}" "$@" -c "$HERE/curve_generic.hip" -o "$TMP/curve_generic.o" &
p5=$!
"$HIPCC" "${FLAGS[@]}" "$@" -c "$HERE/conv_ops.hip" -o "$TMP/conv_ops.o" &
p6=$!
"$HIPCC" "${FLAGS[@]}" "$@" -c "$HERE/critic_step.hip" -o "$TMP/critic_step.o" &
p7=$!
# (a bare `wait` returns 0 whatever the jobs did: wait for each PID so a failed compile stops the script here)
wait $p1
wait $p2
wait $p3
wait $p4
wait $p5
wait $p6
wait $p7
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$TMP/exposure_hip.o" "$TMP/chain_fused.o" "$TMP/nn_ops.o" "$TMP/chain_fused_bwd.o" "$TMP/curve_generic.o" "$TMP/conv_ops.o" "$TMP/critic_step.o" -o "$OUT"
echo "built $OUT (sources $DIGEST)"
