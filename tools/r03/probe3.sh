#!/bin/bash
# r03p3: parity of the statistics' derivative kernels + the training iteration with them in the graph
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p3
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_hip_stats.py tests/test_hip_nets.py tests/test_hip_agent.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest.txt
bash tools/r03/probe2.sh stats
