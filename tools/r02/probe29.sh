#!/bin/bash
# curve backward: clamp in one packed op + integer minima + no "+0" add (18 -> 16 VALU per element pair)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p29
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "curve or filter_matches or golden or ties or every_pixel or reduction or dispatch or chain" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for rep in 1 2 3; do
  EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so timeout 100 python bench.py --no-cpu-baseline > $OUT/head_$rep.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline > $OUT/new_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/head_?.json $OUT/new_?.json | grep -v "cpu\|fwd us"
