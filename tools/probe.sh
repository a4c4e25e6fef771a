#!/bin/bash
# Quick register-pressure probe: compiles three backward kernels only (seconds instead of a minute).
# usage: tools/probe.sh [-D... flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -DEXPO_PROBE "$@" \
  -Rpass-analysis=kernel-resource-usage exposure_amd/csrc/exposure_hip.hip -o /tmp/probe.o 2> /tmp/probe.log
python tools/resource_usage.py /tmp/probe.log | cut -c1-110
