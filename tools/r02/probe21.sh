#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p21
mkdir -p $OUT
for n in 32 128 256 64; do
  timeout 300 python tools/r02/probe20.py $n,512,512 1,2,3 > $OUT/split_$n.txt 2>&1
  echo "== $n"; grep "^[0-9] {" $OUT/split_$n.txt
done
