#!/bin/bash
# r04p22: memory-side counters of the 256x512x512 chain's whole-batch launches (round-3 verdict, item 4: "no counter
# evidence is committed" for the positional outliers): read / write request counts, their queue LEVEL sums (LEVEL / REQ =
# average cycles a request spends outstanding at the fabric interface) and the write / credit stall cycles, per kernel.
# One --pmc pass per counter group (TCC has four slots), --kernel-trace only.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04p22; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
export EXPO_CHAIN_STREAMS=1 EXPO_CHAIN_TILE_MIB=0
Q="--no-cpu-baseline --cold-shape none --no-legs --no-per-kernel --steps 3 --warmup 1"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_BUSY_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  rm -rf /tmp/cc_$i
  rocprofv3 --pmc $set --kernel-trace -d /tmp/cc_$i -o pmc -- python $R/bench.py --shape 256,512,512 $Q > /tmp/cc_$i.log 2>&1
  python $R/tools/rocpd_pmc.py "$(db /tmp/cc_$i)" > $OUT/cold_counters_$i.csv 2>> $OUT/cold_counters.err || echo "pass $i ($set) failed" >> $OUT/cold_counters.err
done
ls -la $OUT
