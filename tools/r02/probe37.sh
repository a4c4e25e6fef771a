#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p37
mkdir -p $OUT
for s in 128,512,512 256,512,512 32,512,512; do
  for rep in 1 2; do
    EXPO_HIP_LIB=$R/tools/r02/libs/curve16.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none --no-per-kernel --shape $s > $OUT/old_s$rep.json 2>/dev/null
    timeout 100 python bench.py --no-cpu-baseline --cold-shape none --no-per-kernel --shape $s > $OUT/new_s$rep.json 2>/dev/null
  done
  echo "== shape $s"; python tools/show_bench.py $OUT/old_s?.json $OUT/new_s?.json | grep "ms/step"
done
