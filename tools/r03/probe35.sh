#!/bin/bash
# r03p35: SQ counters of the one-pass kernels (chain_fused_fwd / chain_fused_bwd), 64x512x512x3 fp16 -- what bounds them?
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p35
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find $1 -name '*.db' | head -1; }
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
P3="SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  timeout 200 rocprofv3 --pmc $P --kernel-trace -d /tmp/pm_$i -o pmc -- python $R/bench.py --workload chain_fused --steps 3 --warmup 1 > $OUT/log_$i.txt 2>&1
  d=$(db /tmp/pm_$i)
  [ -n "$d" ] && python $R/tools/rocpd_pmc.py "$d" | grep -i "chain_fused\|Kernel" > $OUT/pmc_$i.csv
done
cat $OUT/pmc_*.csv
