#!/bin/bash
# Timeline of the training iteration's timed region (gaps between launches).  gpurun -- 'bash tools/r06/train_timeline.sh [tag]'
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-tl}
OUT=$R/gpurun_out/r06_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
STEPS=10
rm -rf /tmp/kt_train
rocprofv3 --kernel-trace -d /tmp/kt_train -o kt -- python $R/bench.py --workload train --steps $STEPS --warmup 3 --graph on > $OUT/bench_train_profiled.json 2> /tmp/kt_train.log
ms=$(python -c "import json; print(json.load(open('$OUT/bench_train_profiled.json'))['ms_per_step'] * $STEPS)")
one=$(python -c "import json; print(json.load(open('$OUT/bench_train_profiled.json'))['ms_per_step'] * 1.0)")
(cd $R/tools && python rocpd_window_stats.py "$(db /tmp/kt_train)" $ms $STEPS) > $OUT/kernel_stats_train.csv
(cd $R/tools && python rocpd_timeline.py "$(db /tmp/kt_train)" $ms 0) > $OUT/timeline_summary.txt
(cd $R/tools && python rocpd_timeline.py "$(db /tmp/kt_train)" $one 600) > $OUT/timeline_one.txt
head -1 $OUT/kernel_stats_train.csv
head -30 $OUT/timeline_summary.txt
