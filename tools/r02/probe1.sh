#!/bin/bash
# round-2 probe 1: MALL-cold evidence + cache-policy threshold sweep (no code changes vs round 1)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p1
mkdir -p $OUT
cd $R
for sz in 96 512 1024; do
  reps=20; [ $sz -ge 512 ] && reps=8
  timeout 120 tools/membench $sz 9 $reps pol > $OUT/membench_${sz}.txt 2>&1
done
timeout 200 python bench.py --shape 256,512,512 --no-cpu-baseline > $OUT/bench_chain_256.json 2> $OUT/bench_chain_256.err
timeout 100 python bench.py --no-cpu-baseline > $OUT/bench_chain_C.json 2> $OUT/bench_chain_C.err
for mb in 8 32 64 128 512; do
  EXPO_STREAM_MIN_BYTES=$((mb<<20)) timeout 100 python bench.py --shape B --no-cpu-baseline > $OUT/bench_chain_B_min${mb}.json 2>/dev/null
  EXPO_STREAM_MIN_BYTES=$((mb<<20)) timeout 100 python bench.py --workload infer --shape B > $OUT/bench_infer_B_min${mb}.json 2>/dev/null
done
for mb in 8 512; do
  EXPO_STREAM_MIN_BYTES=$((mb<<20)) timeout 100 python bench.py --workload infer --shape C > $OUT/bench_infer_C_min${mb}.json 2>/dev/null
done
ls -la $OUT
