#!/bin/bash
# after the clean-up of the compile-time variants: full gpu suite + chain A/B against the previous build
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p26
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for rep in 1 2 3; do
  EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so timeout 100 python bench.py --no-cpu-baseline > $OUT/head_$rep.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline > $OUT/new_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/head_?.json $OUT/new_?.json | grep -v cpu
