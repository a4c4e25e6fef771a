#!/bin/bash
# Training iteration: bench line + kernel table of the timed region (hipGraph replay).  gpurun -- 'bash tools/r06/train_prof.sh [tag]'
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-p}
OUT=$R/gpurun_out/r06_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
python $R/bench.py --workload train --steps 20 --warmup 3 > $OUT/bench_train.json 2> $OUT/bench_train.err
STEPS=10
rm -rf /tmp/kt_train
rocprofv3 --kernel-trace -d /tmp/kt_train -o kt -- python $R/bench.py --workload train --steps $STEPS --warmup 3 --graph on > $OUT/bench_train_profiled.json 2> /tmp/kt_train.log
ms=$(python -c "import json; print(json.load(open('$OUT/bench_train_profiled.json'))['ms_per_step'] * $STEPS)")
(cd $R/tools && python rocpd_window_stats.py "$(db /tmp/kt_train)" $ms $STEPS) > $OUT/kernel_stats_train.csv
python -c "
import json
d=json.load(open('$OUT/bench_train.json'))
print('train: %.3f ms per iteration, frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
"
head -45 $OUT/kernel_stats_train.csv | cut -c1-150
