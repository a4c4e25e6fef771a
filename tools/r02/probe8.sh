#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p8
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_filters.py tests/test_hip_agent.py tests/test_hip_reduction.py -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --cold-shape none"
for rep in 1 2 3; do
  timeout 100 python bench.py $B > $OUT/base_C_$rep.json 2>/dev/null
  EXPO_CHAIN_SNAKE=1 timeout 100 python bench.py $B > $OUT/snake_C_$rep.json 2>/dev/null
  timeout 100 python bench.py $B --shape B > $OUT/base_B_$rep.json 2>/dev/null
  EXPO_CHAIN_SNAKE=1 timeout 100 python bench.py $B --shape B > $OUT/snake_B_$rep.json 2>/dev/null
done
timeout 100 python bench.py $B --shape 256,512,512 > $OUT/base_cold.json 2>/dev/null
EXPO_CHAIN_SNAKE=1 timeout 100 python bench.py $B --shape 256,512,512 > $OUT/snake_cold.json 2>/dev/null
timeout 100 python bench.py $B --shape 128,512,512 > $OUT/base_128.json 2>/dev/null
EXPO_CHAIN_SNAKE=1 timeout 100 python bench.py $B --shape 128,512,512 > $OUT/snake_128.json 2>/dev/null
timeout 100 python bench.py --workload infer --shape B > $OUT/infer_B.json 2>/dev/null
timeout 100 python bench.py --workload infer --shape C > $OUT/infer_C.json 2>/dev/null
