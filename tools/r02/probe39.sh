#!/bin/bash
# curve backward: slope table staged from a per-lane parameter copy through shuffles (no global loads behind the
# first image chunk) vs the committed staging
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p39
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "curve or filter_matches or golden or ties or every_pixel or reduction or chain or dispatch or exceed" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for rep in 1 2; do
  for cfg in "2 4" "1 2" "2 2"; do
    set -- $cfg
    EXPO_TONE_GROUPS_PER_THREAD=$1 EXPO_COLOR_GROUPS_PER_THREAD=$2 EXPO_HIP_LIB=$R/tools/r02/libs/geom.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/old_t$1c$2_$rep.json 2>/dev/null
    EXPO_TONE_GROUPS_PER_THREAD=$1 EXPO_COLOR_GROUPS_PER_THREAD=$2 timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/new_t$1c$2_$rep.json 2>/dev/null
  done
done
python tools/show_bench.py $OUT/old_t*.json $OUT/new_t*.json | grep -v "cpu\|fwd us\|roofline"
