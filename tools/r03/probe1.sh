#!/bin/bash
# r03p1: BEFORE numbers for the training iteration (config 3): bench line, kernel table (eager and hipGraph replay)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
python $R/bench.py --workload train --steps 10 --warmup 3 > $OUT/bench_train_graph.json 2> $OUT/bench_train_graph.err
python $R/bench.py --workload train --steps 10 --warmup 3 --graph off > $OUT/bench_train_eager.json 2> $OUT/bench_train_eager.err
for mode in on off; do
  rm -rf /tmp/kt_$mode
  timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$mode -o kt -- python $R/bench.py --workload train --steps 5 --warmup 3 --graph $mode > /tmp/kt_$mode.log 2>&1
  python $R/tools/rocpd_stats.py "$(db /tmp/kt_$mode)" > $OUT/kernel_stats_train_$mode.csv
done
cat $OUT/bench_train_graph.json $OUT/bench_train_eager.json
head -30 $OUT/kernel_stats_train_off.csv | cut -c1-200
