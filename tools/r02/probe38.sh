#!/bin/bash
# how much of the curve backward is the per-block slope-table staging?  diagnostic build with a trivial table fill
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p38
mkdir -p $OUT
for g in 1 2 4; do
  EXPO_TONE_GROUPS_PER_THREAD=$g EXPO_COLOR_GROUPS_PER_THREAD=$g timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/base_g$g.json 2>/dev/null
  EXPO_TONE_GROUPS_PER_THREAD=$g EXPO_COLOR_GROUPS_PER_THREAD=$g EXPO_HIP_LIB=$R/tools/r02/libs/dbg_stage.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/stage_g$g.json 2>/dev/null
done
python tools/show_bench.py $OUT/base_g?.json $OUT/stage_g?.json | grep -v "cpu\|fwd us\|roofline"
