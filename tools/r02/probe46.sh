#!/bin/bash
# the 26 vs 29 us copy skeleton: L2 <-> fabric traffic of the fast and the slow variants
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p46
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find $1 -name '*.db' | head -1; }
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_LEVEL_sum TCC_EA_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pm_$n
  timeout 120 rocprofv3 --pmc $c --kernel-trace -d /tmp/pm_$n -o pmc -- $R/tools/membench 96 9 6 work > $OUT/log_$n.txt 2>&1
  d=$(db /tmp/pm_$n)
  [ -n "$d" ] && python $R/tools/rocpd_pmc.py "$d" | grep "cpol<2, 16>\|cpol3<5, 0>\|cpol3<6, 0>\|cpol3<0, 0>\|Kernel" > $OUT/pmc_$n.csv
  cat $OUT/pmc_$n.csv | cut -c1-160
done
grep "reference\|kind[56] \|kind0 count0" $OUT/log_FETCH_SIZE.txt
