#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p3
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_reduction.py -x -q > $OUT/pytest_reduction.txt 2>&1
tail -15 $OUT/pytest_reduction.txt
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -8 $OUT/pytest_gpu.txt
B="--no-cpu-baseline --cold-shape none"
for rep in 1 2; do
  (cd tools/r02/old && timeout 100 python bench.py --no-cpu-baseline > $OUT/old_C_$rep.json 2>/dev/null)
  timeout 100 python bench.py $B > $OUT/new_C_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$R/tools/r02/libs/map2.so timeout 100 python bench.py $B > $OUT/map2_C_$rep.json 2>/dev/null
  for g in 1 2 8; do
    EXPO_BWD_GROUPS_PER_THREAD=$g timeout 100 python bench.py $B > $OUT/new_C_g${g}_$rep.json 2>/dev/null
    EXPO_BWD_GROUPS_PER_THREAD=$g EXPO_HIP_LIB=$R/tools/r02/libs/map2.so timeout 100 python bench.py $B > $OUT/map2_C_g${g}_$rep.json 2>/dev/null
  done
  (cd tools/r02/old && timeout 100 python bench.py --no-cpu-baseline --shape B > $OUT/old_B_$rep.json 2>/dev/null)
  timeout 100 python bench.py $B --shape B > $OUT/new_B_$rep.json 2>/dev/null
  (cd tools/r02/old && timeout 100 python bench.py --no-cpu-baseline --shape A > $OUT/old_A_$rep.json 2>/dev/null)
  timeout 100 python bench.py $B --shape A > $OUT/new_A_$rep.json 2>/dev/null
done
timeout 200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -5 $OUT/bench_default.err
