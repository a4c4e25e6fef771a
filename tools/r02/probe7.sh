#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p7
mkdir -p $OUT
cd $R
rm -rf gpurun_out/r02soak
bash tools/r02/soak_rccl.sh 24 fixed
cat gpurun_out/r02soak/summary.txt
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
