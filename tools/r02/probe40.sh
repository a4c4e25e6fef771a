#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p40
mkdir -p $OUT
for rep in 1 2; do
  EXPO_HIP_LIB=$R/tools/r02/libs/geom.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/old_C_$rep.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/new_C_$rep.json 2>/dev/null
  EXPO_TONE_GROUPS_PER_THREAD=1 EXPO_COLOR_GROUPS_PER_THREAD=1 timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/new11_C_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/old_C_?.json $OUT/new_C_?.json $OUT/new11_C_?.json | grep -v "cpu\|fwd us\|roofline"
for s in A B 256,512,512; do
  for rep in 1 2; do
    EXPO_HIP_LIB=$R/tools/r02/libs/geom.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none --shape $s > $OUT/old_s$rep.json 2>/dev/null
    timeout 100 python bench.py --no-cpu-baseline --cold-shape none --shape $s > $OUT/new_s$rep.json 2>/dev/null
  done
  echo "== shape $s"; python tools/show_bench.py $OUT/old_s?.json $OUT/new_s?.json | grep "ms/step\|bwd us"
done
