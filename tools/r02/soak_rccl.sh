#!/bin/bash
# soak of the hipGraph-captured RCCL training step on ONE GPU: where and how does the ~1/15 abort happen?
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02soak
mkdir -p $OUT
cd $R
export EXPO_FORCE_COLLECTIVES=1 HSA_ENABLE_IPC_MODE_LEGACY=0 EXPO_TRACE=1
N=${1:-16}
run_variant() {  # name, extra env...
  name=$1; shift
  fails=0
  for i in $(seq 1 $N); do
    env "$@" timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port $((29600 + i)) bench.py --gpus 1 --workload train --steps 3 --warmup 2 > $OUT/${name}_$i.out 2> $OUT/${name}_$i.err
    rc=$?
    if [ $rc -ne 0 ]; then
      fails=$((fails+1)); echo "$name run $i rc=$rc"; grep -v "^$" $OUT/${name}_$i.err | tail -25 > $OUT/${name}_fail_$i.txt
    else
      rm -f $OUT/${name}_$i.err $OUT/${name}_$i.out
    fi
  done
  echo "$name: $fails / $N failed" | tee -a $OUT/summary.txt
}
run_variant ${2:-fixed}
