#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p10
rm -rf $OUT; mkdir -p $OUT
cd $R
export AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 MIOPEN_LOG_LEVEL=5 MIOPEN_ENABLE_LOGGING_CMD=1
timeout 1500 python -X faulthandler -m pytest tests/test_hip_filters.py tests/test_hip_nets.py -x -q -s -m gpu > /tmp/full.txt 2>&1
echo "rc=$?"
grep -v "Extension modules\|^  File" /tmp/full.txt | tail -120 | cut -c1-400 > $OUT/tail.txt
cat $OUT/tail.txt
