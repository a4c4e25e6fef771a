#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p9
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python tools/show_bench.py $OUT/bench_default.json | head -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline --cold-shape none > /tmp/kt.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | head -1)" > $OUT/kernel_stats.csv
grep -i "finish\|bwd_kernelINS_6CurveFILi3\|ExposureF" $OUT/kernel_stats.csv
