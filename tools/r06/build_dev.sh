#!/bin/bash
# Development build: object files cached under /tmp/expo_obj, only the translation units named on the command line are
# recompiled (default: conv_ops), extra flags after "--".  The binary's digest is stale: run probes with EXPO_ALLOW_STALE_LIB=1
# and finish with exposure_amd/csrc/build.sh.   usage: bash tools/r06/build_dev.sh [unit ...] [-- -DFLAG ...]
set -euo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
C=$R/exposure_amd/csrc
O=/tmp/expo_obj
mkdir -p $O
units=(); extra=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi
  units+=("$1"); shift
done
[ ${#units[@]} -eq 0 ] && units=(conv_ops)
ALL=(exposure_hip chain_fused nn_ops chain_fused_bwd curve_generic conv_ops critic_step)
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC)
pids=()
for u in "${ALL[@]}"; do
  need=0
  [ -f $O/$u.o ] || need=1
  for v in "${units[@]}"; do [ "$v" == "$u" ] && need=1; done
  if [ $need -eq 1 ]; then
    f=("${FLAGS[@]}")
    [ $u == chain_fused ] && f+=(-fno-slp-vectorize -fno-honor-nans)
    [ $u == chain_fused_bwd ] && f+=(-fno-slp-vectorize)
    /opt/rocm/bin/hipcc "${f[@]}" "${extra[@]}" -c $C/$u.hip -o $O/$u.o 2>$O/$u.log &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p || { grep -h -A3 "error" $O/*.log | head -30; exit 1; }; done
objs=(); for u in "${ALL[@]}"; do objs+=($O/$u.o); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $R/exposure_amd/libexposure_hip.so
echo "dev build done (${units[*]}; ${extra[*]:-no extra flags})"
