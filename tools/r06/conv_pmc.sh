#!/bin/bash
# per-dispatch durations and SQ counters of every convolution launch shape: gpurun -- 'bash tools/r06/conv_pmc.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r06_convpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp_kt /tmp/cp_pmc1 /tmp/cp_pmc2
rocprofv3 --kernel-trace --output-format csv -d /tmp/cp_kt -o kt -- python $R/tools/r06/conv_cases.py 10 > /tmp/cp_kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/cp_pmc1 -o p1 -- python $R/tools/r06/conv_cases.py 3 > /tmp/cp_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/cp_pmc2 -o p2 -- python $R/tools/r06/conv_cases.py 3 > /tmp/cp_pmc2.log 2>&1
find /tmp/cp_kt /tmp/cp_pmc1 /tmp/cp_pmc2 -name "*.csv" | head -20
for f in $(find /tmp/cp_kt -name "*kernel_trace.csv"); do cp $f $OUT/kt_kernel_trace.csv; done
for f in $(find /tmp/cp_pmc1 -name "*counter_collection.csv"); do cp $f $OUT/p1_counters.csv; done
for f in $(find /tmp/cp_pmc2 -name "*counter_collection.csv"); do cp $f $OUT/p2_counters.csv; done
ls -la $OUT; tail -3 /tmp/cp_pmc1.log
