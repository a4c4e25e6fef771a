#!/bin/bash
# forward Tone / Color: the table's per-lane parameter fetched before the image loads
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p41
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "curve or filter_matches or golden or every_pixel or chain or dispatch or fused" > $OUT/pytest.txt 2>&1
tail -2 $OUT/pytest.txt
for rep in 1 2 3; do
  EXPO_HIP_LIB=$R/tools/r02/libs/stage_lanes.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/old_C_$rep.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/new_C_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/old_C_?.json $OUT/new_C_?.json | grep -v "cpu\|bwd us\|roofline"
for rep in 1 2; do
  EXPO_HIP_LIB=$R/tools/r02/libs/stage_lanes.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none --shape B > $OUT/old_B_$rep.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline --cold-shape none --shape B > $OUT/new_B_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/old_B_?.json $OUT/new_B_?.json | grep -v "cpu\|bwd us\|roofline"
