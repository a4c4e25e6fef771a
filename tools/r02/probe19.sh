#!/bin/bash
# what bounds the fused inference kernel: SQ counters, scalar vs pixel-pair build
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=$R/gpurun_out/r02p19
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find $1 -name '*.db' | head -1; }
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
for lib in pair scalar; do
  [ $lib = scalar ] && export EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so || unset EXPO_HIP_LIB
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rm -rf /tmp/pm_${lib}_$i
    timeout 200 rocprofv3 --pmc $P --kernel-trace -d /tmp/pm_${lib}_$i -o pmc -- python $R/bench.py --workload infer --shape B --steps 5 --warmup 2 > $OUT/log_${lib}_$i.txt 2>&1
    d=$(db /tmp/pm_${lib}_$i)
    [ -n "$d" ] && python $R/tools/rocpd_pmc.py "$d" | grep -i "fused\|Kernel" > $OUT/pmc_${lib}_$i.csv
  done
done
cat $OUT/pmc_*.csv | cut -c1-30,150-400
