#!/bin/bash
# upper bound of a free block epilogue: diagnostic build without the block reduction at 1 / 2 / 4 groups per thread
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p35
mkdir -p $OUT
for g in 1 2 4; do
  EXPO_BWD_GROUPS_PER_THREAD=$g timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/base_g$g.json 2>/dev/null
  EXPO_BWD_GROUPS_PER_THREAD=$g EXPO_HIP_LIB=$R/tools/r02/libs/dbg_noepi.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/noepi_g$g.json 2>/dev/null
  EXPO_BWD_GROUPS_PER_THREAD=$g EXPO_HIP_LIB=$R/tools/r02/libs/dbg_trivial.so timeout 100 python bench.py --no-cpu-baseline --cold-shape none > $OUT/trivial_g$g.json 2>/dev/null
done
python tools/show_bench.py $OUT/base_g?.json $OUT/noepi_g?.json $OUT/trivial_g?.json | grep -v "cpu\|fwd us\|roofline"
