#!/bin/bash
# forward kernels: conversion-only diagnostic build vs the real kernels vs the copy skeleton
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p42
mkdir -p $OUT
for rep in 1 2; do
  timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/base_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$R/tools/r02/libs/dbg_fwd.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/trivial_$rep.json 2>/dev/null
  EXPO_FWD_GROUPS_PER_THREAD=2 timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/base_g2_$rep.json 2>/dev/null
done
python tools/show_bench.py $OUT/base_?.json $OUT/trivial_?.json $OUT/base_g2_?.json | grep -v "cpu\|bwd us\|roofline"
tools/membench 96 9 20 pol 2>/dev/null | grep -i "cpol\|copy" | tail -12
