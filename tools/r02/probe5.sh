#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p5
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
B="--no-cpu-baseline --cold-shape none"
L=$R/tools/r02/libs
for rep in 1 2; do
  (cd tools/r02/old && timeout 100 python bench.py --no-cpu-baseline > $OUT/old_C_$rep.json 2>/dev/null)
  timeout 100 python bench.py $B > $OUT/base_C_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$L/pingpong.so timeout 100 python bench.py $B > $OUT/pingpong_C_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$L/noslp.so timeout 100 python bench.py $B > $OUT/noslp_C_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$L/fwdpf.so EXPO_FWD_GROUPS_PER_THREAD=2 timeout 100 python bench.py $B > $OUT/fwdpf2_C_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$L/fwdpf.so EXPO_FWD_GROUPS_PER_THREAD=4 timeout 100 python bench.py $B > $OUT/fwdpf4_C_$rep.json 2>/dev/null
  EXPO_FWD_GROUPS_PER_THREAD=2 timeout 100 python bench.py $B > $OUT/fwdg2_C_$rep.json 2>/dev/null
  timeout 100 python bench.py $B --shape B > $OUT/base_B_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$L/pingpong.so timeout 100 python bench.py $B --shape B > $OUT/pingpong_B_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$L/noslp.so timeout 100 python bench.py $B --shape B > $OUT/noslp_B_$rep.json 2>/dev/null
done
for g in 4 8 16; do
  EXPO_RED_GROUPS_PER_THREAD=$g timeout 100 python tools/bench_extra.py > $OUT/extra_r${g}.json 2>/dev/null
done
timeout 100 python bench.py --workload infer --shape B > $OUT/infer_B.json 2>/dev/null
EXPO_HIP_LIB=$L/noslp.so timeout 100 python bench.py --workload infer --shape B > $OUT/infer_B_noslp.json 2>/dev/null
