#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p30
mkdir -p $OUT
for rep in 1 2; do
  for g in 4 8 16; do
    EXPO_BWD_GROUPS_PER_THREAD=$g timeout 100 python bench.py --no-cpu-baseline > $OUT/gpt${g}_$rep.json 2>/dev/null
  done
done
python tools/show_bench.py $OUT/gpt*_?.json | grep -v "cpu\|fwd us"
