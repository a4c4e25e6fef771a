#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
rm -rf gpurun_out/r02soak
bash tools/r02/soak_rccl.sh 36 fixed2
cat gpurun_out/r02soak/summary.txt
