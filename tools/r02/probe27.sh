#!/bin/bash
# element-wise backward: 1 / 2 / 4 partial accumulators per thread
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p27
mkdir -p $OUT
for rep in 1 2 3; do
  EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so timeout 100 python bench.py --no-cpu-baseline > $OUT/head_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$R/tools/r02/libs/acc1.so timeout 100 python bench.py --no-cpu-baseline > $OUT/acc1_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$R/tools/r02/libs/acc2.so timeout 100 python bench.py --no-cpu-baseline > $OUT/acc2_$rep.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline > $OUT/acc4_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/head_?.json $OUT/acc1_?.json $OUT/acc2_?.json $OUT/acc4_?.json | grep -v "cpu\|fwd us"
