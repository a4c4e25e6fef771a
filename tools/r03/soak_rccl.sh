#!/bin/bash
# r03 soak of the hipGraph-captured RCCL training step on ONE GPU with the VERIFIED watchdog drain
# (exposure_amd.dist.drain_before_capture): N runs, each must end rc 0 with capture_drain_verified = true.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03soak
rm -rf $OUT; mkdir -p $OUT
cd $R
export EXPO_FORCE_COLLECTIVES=1 HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-20}
fails=0; unverified=0
for i in $(seq 1 $N); do
  timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $((29700 + i)) bench.py --gpus 1 --workload train --steps 3 --warmup 2 > $OUT/run_$i.out 2> $OUT/run_$i.err
  rc=$?
  if [ $rc -ne 0 ]; then
    fails=$((fails+1)); echo "run $i rc=$rc"; grep -v "^$" $OUT/run_$i.err | tail -25 > $OUT/fail_$i.txt
  else
    python - <<PY
import json
line = [l for l in open('$OUT/run_$i.out') if l.startswith('{')][-1]
c = json.loads(line)['config']
print('run $i: %s, drain verified: %s' % (c['launch'], c['capture_drain_verified']))
PY
    grep -q '"capture_drain_verified": true' $OUT/run_$i.out || unverified=$((unverified+1))
    rm -f $OUT/run_$i.err
  fi
done
echo "aborted: $fails / $N; drain not verified: $unverified / $N" | tee $OUT/summary.txt
