#!/bin/bash
# r03p2: steady-state kernel table of the training iteration (config 3) -- the timed region only
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-before}
OUT=$R/gpurun_out/r03p2_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
STEPS=10
for mode in on off; do
  rm -rf /tmp/kt_$mode
  timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$mode -o kt -- python $R/bench.py --workload train --steps $STEPS --warmup 3 --graph $mode > $OUT/bench_train_$mode.json 2> /tmp/kt_$mode.log
  ms=$(python -c "import json,sys; print(json.load(open('$OUT/bench_train_$mode.json'))['ms_per_step'] * $STEPS)")
  (cd $R/tools && python rocpd_window_stats.py "$(db /tmp/kt_$mode)" $ms $STEPS) > $OUT/kernel_stats_train_$mode.csv
  head -1 $OUT/kernel_stats_train_$mode.csv
done
python $R/bench.py --workload train --steps 20 --warmup 3 > $OUT/bench_train.json 2>/dev/null
cat $OUT/bench_train.json | cut -c1-300
head -42 $OUT/kernel_stats_train_on.csv | cut -c1-180
