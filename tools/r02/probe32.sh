#!/bin/bash
# curve backward with the two-deep prefetch (occupancy 3, two chunks in flight) vs one-deep (occupancy 4)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p32
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "curve or filter_matches or golden or ties or every_pixel or reduction or chain or ragged or huge or sweep" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for rep in 1 2 3; do
  EXPO_HIP_LIB=$R/tools/r02/libs/curve16.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/one_$rep.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/deep_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/one_?.json $OUT/deep_?.json | grep -v "cpu\|fwd us"
for s in B 256,512,512; do
  EXPO_HIP_LIB=$R/tools/r02/libs/curve16.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none --shape $s > $OUT/one_s.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline --cold-shape none --shape $s > $OUT/deep_s.json 2>/dev/null
  python tools/show_bench.py $OUT/one_s.json $OUT/deep_s.json | grep -v "cpu\|fwd us"
done
