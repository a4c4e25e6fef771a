#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p12
rm -rf $OUT; mkdir -p $OUT
cd $R
for find in 0 1; do
  EXPO_MIOPEN_FIND=$find python bench.py --workload train --steps 20 --warmup 3 2>/dev/null | cut -c1-260 | tee $OUT/train_find$find.json
done
EXPO_MIOPEN_FIND=0 timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu > $OUT/pytest_find0.txt 2>&1
echo "find0 suite rc=$?"; tail -3 $OUT/pytest_find0.txt | cut -c1-200
