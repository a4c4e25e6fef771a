#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p2
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 python bench.py --workload train --no-cpu-baseline > $OUT/bench_train.json 2> $OUT/bench_train.err
tail -c 600 $OUT/bench_default.json
