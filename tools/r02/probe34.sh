#!/bin/bash
# where do the ~4 us between the 2-read-1-write skeleton and the light backward kernels go?
# diagnostic builds (results are wrong by construction; timing only): no block reduction / trivial arithmetic / both
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p34
mkdir -p $OUT
for rep in 1 2; do
  timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/base_$rep.json 2>/dev/null
  for v in noepi trivial trivial_epi; do
    EXPO_HIP_LIB=$R/tools/r02/libs/dbg_$v.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/${v}_$rep.json 2>/dev/null
  done
done
python tools/show_bench.py $OUT/base_?.json $OUT/noepi_?.json $OUT/trivial_epi_?.json $OUT/trivial_?.json | grep -v "cpu\|fwd us\|roofline"
tools/membench 96 9 20 pol 2>/dev/null | tail -8
