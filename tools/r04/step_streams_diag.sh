#!/bin/bash
# r04p23 (diagnostic, run on the tree of r04p16 with tools/r04/step_streams.patch applied): where do the 2 ms go when the
# critic step's two branches run on two streams?  rocprofv3 window statistics of the timed region for both settings.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04p23; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
STEPS=10
for tag in one critic2; do
  if [ $tag = one ]; then export EXPO_STEP_STREAMS=1; else unset EXPO_STEP_STREAMS; export EXPO_STEP_STREAMS_WHERE=c; fi
  rm -rf /tmp/kt_$tag
  rocprofv3 --kernel-trace -d /tmp/kt_$tag -o kt -- python $R/bench.py --workload train --steps $STEPS --warmup 3 > $OUT/bench_$tag.json 2> /tmp/kt_$tag.log
  ms=$(python -c "import json; print(json.load(open('$OUT/bench_$tag.json'))['ms_per_step'] * $STEPS)")
  (cd $R/tools && python rocpd_window_stats.py "$(db /tmp/kt_$tag)" $ms $STEPS) > $OUT/window_$tag.csv
  head -1 $OUT/window_$tag.csv
  python $R/bench.py --workload train --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; print('$tag unprofiled: %.3f ms' % json.loads(sys.stdin.read())['ms_per_step'])"
done
