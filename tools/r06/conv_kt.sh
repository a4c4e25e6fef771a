#!/bin/bash
# per-shape durations of every convolution launch (rocprofv3 --kernel-trace): gpurun -- 'bash tools/r06/conv_kt.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r06_convkt
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp_kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/cp_kt -o kt -- python $R/tools/r06/conv_cases.py 10 > /tmp/cp_kt.log 2>&1
for f in $(find /tmp/cp_kt -name "*kernel_trace.csv"); do cp $f $OUT/kt_kernel_trace.csv; done
python $R/tools/r06/conv_table.py $OUT/kt_kernel_trace.csv | tee $OUT/conv_table.txt
