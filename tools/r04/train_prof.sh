R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04p15; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
STEPS=10
rm -rf /tmp/kt_train_on
rocprofv3 --kernel-trace -d /tmp/kt_train_on -o kt -- python $R/bench.py --workload train --steps $STEPS --warmup 3 --graph on > $OUT/bench_train_profiled_on.json 2> /tmp/kt_train_on.log
ms=$(python -c "import json; print(json.load(open('$OUT/bench_train_profiled_on.json'))['ms_per_step'] * $STEPS)")
(cd $R/tools && python rocpd_window_stats.py "$(db /tmp/kt_train_on)" $ms $STEPS) > $OUT/kernel_stats_train_graph_on.csv
python $R/bench.py --workload train --steps 20 --warmup 3 > $OUT/bench_train.json 2>/dev/null
head -c 600 $OUT/bench_train.json
