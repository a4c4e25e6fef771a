#!/bin/bash
# r03p32: which MIOpen solvers the training iteration's convolutions take (immediate mode) -- switch families off
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() {
  echo "== $1"
  shift
  env "$@" timeout 300 python bench.py --workload train --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('kernel_launches_per_iteration', d['config'].get('launches')))"
}
for rep in 1 2; do
run "baseline" X=1
run "wrw gtc xdlops nhwc off" MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0
run "bwd gtc xdlops nhwc off" MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0
run "fwd gtc xdlops nhwc off" MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0
run "all three off" MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0
done
