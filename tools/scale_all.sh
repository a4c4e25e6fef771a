#!/bin/bash
# The whole SURVEY.md section 8(e) scaling table in ONE command on a node with 8 MI355X:
#   bash tools/scale_all.sh [out_dir]            (about 6 minutes)
# rows: {filter chain, training iteration, gradient all-reduce only} x {weak, strong} x N = 1, 2, 4, 8.
# bench.py --gpus N starts its N ranks itself (torch.distributed.run, rendezvous on 127.0.0.1, one process per
# GPU over RCCL); every run prints one JSON line with n_gpus = N; tools/scale_table.py turns them into the table
# (scaling efficiency = value(N) / (N * value(1)) for weak, value(N) / value(1) / N for strong).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
OUT=${1:-$R/gpurun_out/scale}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TORCH_FR_BUFFER_SIZE=${TORCH_FR_BUFFER_SIZE:-2000}
NG=$(python -c "import torch; print(torch.cuda.device_count())")
# rehearsal on a 1-GPU box: EXPO_BENCH_SHARE_GPU=1 lets the ranks share the GPU over gloo (bench.py); the table then
# shows the plumbing works, its numbers mean nothing
[ "${EXPO_BENCH_SHARE_GPU:-0}" = 1 ] && NG=${EXPO_SCALE_MAX_RANKS:-2}
Q="--no-cpu-baseline --no-per-kernel --cold-shape none"
for n in 1 2 4 8; do
  [ $n -gt $NG ] && break
  for sc in weak strong; do
    python $R/bench.py --gpus $n --scaling $sc $Q > $OUT/chain_${sc}_$n.json 2> $OUT/chain_${sc}_$n.err
    python $R/bench.py --gpus $n --scaling $sc --workload train --steps 20 --warmup 3 > $OUT/train_${sc}_$n.json 2> $OUT/train_${sc}_$n.err
    # the same iteration with eager (un-captured) launches: what the hipGraph replay of the steps is worth at this N
    python $R/bench.py --gpus $n --scaling $sc --workload train --steps 10 --warmup 3 --graph off > $OUT/train_eager_${sc}_$n.json 2> $OUT/train_eager_${sc}_$n.err
  done
  python $R/bench.py --gpus $n --workload allreduce --steps 50 --warmup 5 > $OUT/allreduce_$n.json 2> $OUT/allreduce_$n.err
done
python $R/tools/scale_table.py $OUT | tee $OUT/scale_table.md
