#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p6
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_agent.py tests/test_hip_nets.py tests/test_hip_evaluate.py -x -q > $OUT/pytest_agent.txt 2>&1
tail -5 $OUT/pytest_agent.txt
timeout 100 python bench.py --workload train --no-cpu-baseline > $OUT/train_graph.json 2> $OUT/train_graph.err
timeout 100 python bench.py --workload train --graph off --no-cpu-baseline > $OUT/train_eager.json 2> $OUT/train_eager.err
timeout 100 python tools/bench_extra.py > $OUT/extra.json 2>/dev/null
bash tools/r02/soak_rccl.sh 14
cat gpurun_out/r02soak/summary.txt
