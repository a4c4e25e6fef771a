#!/bin/bash
# per-filter groups per thread for the backward kernels (light 1, Tone 2, Color 4) vs 4 for all
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p36
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for rep in 1 2 3; do
  EXPO_HIP_LIB=$R/tools/r02/libs/curve16.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/old_C_$rep.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/new_C_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/old_C_?.json $OUT/new_C_?.json | grep -v "cpu\|fwd us\|roofline"
for s in A B 256,512,512; do
  for rep in 1 2; do
    EXPO_HIP_LIB=$R/tools/r02/libs/curve16.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none --shape $s > $OUT/old_s$rep.json 2>/dev/null
    timeout 100 python bench.py --no-cpu-baseline --cold-shape none --shape $s > $OUT/new_s$rep.json 2>/dev/null
  done
  echo "== shape $s"; python tools/show_bench.py $OUT/old_s?.json $OUT/new_s?.json | grep "ms/step"
done
timeout 100 python bench.py --no-cpu-baseline --cold-shape none --dtype f32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 new', d['ms_per_step'])"
EXPO_HIP_LIB=$R/tools/r02/libs/curve16.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none --dtype f32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 old', d['ms_per_step'])"
