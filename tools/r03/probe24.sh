#!/bin/bash
# r03p24: training iteration after dropping the parameter gradients autograd discards (GP inner grad, G-step critic/value passes)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p24
rm -rf $OUT; mkdir -p $OUT
cd $R
for rep in 1 2; do
for find in off on; do
  python bench.py --workload train --steps 20 --warmup 3 --miopen-find $find 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('find $find rep $rep: %.3f ms  %.0f img/s' % (d['ms_per_step'], d['value']))"
done
done
timeout 900 python -m pytest tests/test_hip_agent.py tests/test_hip_nets.py tests/test_hip_stats.py tests/test_nn_ops.py -x -q -m gpu 2>&1 | tail -3
bash tools/r03/probe2.sh skipgrads 2>&1 | head -3
