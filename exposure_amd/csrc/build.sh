#!/bin/bash
# Build libexposure_hip.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libexposure_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" \
  "$HERE/exposure_hip.hip" -o "$OUT"
echo "built $OUT"
