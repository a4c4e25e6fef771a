#!/bin/bash
# Regenerates every measured artefact under profiles/ in ONE gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/profile_all.sh'
# then, back in the container:  tools/collect_profiles.sh r06_final
# (an 8-GPU lease additionally runs tools/scale_all.sh: the section 8(e) scaling table)
# rocprofv3 passes: --kernel-trace alone (durations) and one --pmc counter per pass (never combined
# with sys/runtime tracing); the rocpd databases stay on the GPU box, only CSV summaries come back.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
COLD=256,512,512
Q="--no-cpu-baseline --cold-shape none --no-legs"

# the gpu suite, recording the worst |err| / A of every reduced-quantity comparison (tests/_tol.py)
EXPO_RECORD_PARAM_ERR=$OUT/param_grad_errors.jsonl python -m pytest $R/tests -x -q -m gpu 2>&1 | tail -2 > $OUT/pytest_gpu.txt
python $R/bench.py > $OUT/bench_chain.json 2> $OUT/bench_chain.err
EXPO_CHAIN_STREAMS=1 python $R/bench.py $Q > $OUT/bench_chain_1stream.json 2>/dev/null
python $R/bench.py --shape A $Q > $OUT/bench_chain_A.json 2>/dev/null
python $R/bench.py --shape B $Q > $OUT/bench_chain_B.json 2>/dev/null
python $R/bench.py --dtype f32 $Q > $OUT/bench_chain_f32.json 2>/dev/null
python $R/bench.py --shape $COLD $Q > $OUT/bench_chain_cold.json 2>/dev/null  # chain calls run tile-major (round 4)
EXPO_CHAIN_TILE_MIB=0 python $R/bench.py --shape $COLD $Q > $OUT/bench_chain_cold_untiled.json 2>/dev/null  # round 3's alternating walk
python $R/bench.py --workload infer --shape B > $OUT/bench_infer_B.json 2>/dev/null
python $R/bench.py --workload infer --shape C > $OUT/bench_infer_C.json 2>/dev/null
python $R/bench.py --workload chain_fused > $OUT/bench_chain_fused.json 2>/dev/null
python $R/bench.py --workload chain_fused --shape B > $OUT/bench_chain_fused_B.json 2>/dev/null
python $R/bench.py --workload chain_fused --dtype f32 --shape B > $OUT/bench_chain_fused_f32_B.json 2>/dev/null
python $R/bench.py --workload train --steps 40 --warmup 6 > $OUT/bench_train.json 2>/dev/null  # (the iteration graph's first replays are slower: six untimed iterations)
python $R/bench.py --workload train --steps 10 --warmup 3 --graph off > $OUT/bench_train_eager.json 2>/dev/null
python $R/tools/r06/iter_probe.py 16 2>/dev/null | grep '^it' > $OUT/iter_probe.txt  # host time per iteration, planned or step-by-step
python $R/tools/bench_extra.py > $OUT/bench_extra.json 2>/dev/null
python $R/tools/r05/conv_bench.py all --reps 20 > $OUT/conv_bench.txt 2>&1  # the in-house convolution kernels against MIOpen (its own heuristics: no rankings are shipped any more), per layer
python $R/tools/r06/conv_sweep.py all 2>/dev/null | grep -v amdgpu.ids > $OUT/conv_sweep.txt  # every decomposition at batch 64 / 192
python $R/tools/r06/step_bench.py 2>/dev/null | grep "step" > $OUT/step_bench.txt  # critic update and G / V step, hand-scheduled vs autograd
python $R/tools/r06/step_trace.py gc 2>/dev/null > $OUT/step_trace.txt  # the launches of one eager G / V step and one critic update, in order
for sz in 24 96 512 1024; do  # (24 MiB = BASELINE config 5's tensors: the ceiling of the per-step chain at 16 x 512 x 512)
  reps=20; [ $sz -ge 512 ] && reps=8
  $R/tools/membench $sz 9 $reps pol > $OUT/membench_${sz}.txt 2>&1
done
$R/tools/membench > $OUT/membench.txt 2>&1

# ---- kernel durations (rocprofv3 --kernel-trace), one table per workload
kt() {  # name, command...
  name=$1; shift
  rm -rf /tmp/kt_$name
  rocprofv3 --kernel-trace -d /tmp/kt_$name -o kt -- "$@" > /tmp/kt_$name.log 2>&1
  python $R/tools/rocpd_stats.py "$(db /tmp/kt_$name)" > $OUT/kernel_stats_$name.csv
}
# the chain: ONE run of the default command; its whole-batch launches (grid_y = 64: bench.py's per-kernel leg, what
# roofline.* is quoted on) and the overlapping half-batch launches of the timed chain (grid_y = 32) in separate tables
kt chain_all python $R/bench.py $Q
python $R/tools/rocpd_stats.py "$(db /tmp/kt_chain_all)" grid_y=64,8 > $OUT/kernel_stats_chain.csv  # (8: the finish launch, grid (images, steps))
python $R/tools/rocpd_stats.py "$(db /tmp/kt_chain_all)" grid_y=32 > $OUT/kernel_stats_chain_halves.csv
kt cold_all python $R/bench.py --shape $COLD $Q
python $R/tools/rocpd_stats.py "$(db /tmp/kt_cold_all)" grid_y=256,8 > $OUT/kernel_stats_cold.csv  # whole-batch launches of the per-kernel leg
python $R/tools/rocpd_stats.py "$(db /tmp/kt_cold_all)" grid_y=32 > $OUT/kernel_stats_cold_tiles.csv  # the chain calls: half-tile launches
kt chain_A python $R/bench.py --shape A $Q
kt chain_B python $R/bench.py --shape B $Q
kt infer_B python $R/bench.py --workload infer --shape B
kt infer_C python $R/bench.py --workload infer --shape C
kt extra python $R/tools/bench_extra.py
kt chain_fused python $R/bench.py --workload chain_fused
cp $OUT/kernel_stats_chain.csv $OUT/kernel_stats.csv
# the training iteration (BASELINE config 3): the timed region only (tools/rocpd_window_stats.py)
STEPS=10
for mode in on off; do
  rm -rf /tmp/kt_train_$mode
  rocprofv3 --kernel-trace -d /tmp/kt_train_$mode -o kt -- python $R/bench.py --workload train --steps $STEPS --warmup 6 --graph $mode > $OUT/bench_train_profiled_$mode.json 2> /tmp/kt_train_$mode.log
  ms=$(python -c "import json; print(json.load(open('$OUT/bench_train_profiled_$mode.json'))['ms_per_step'] * $STEPS)")
  (cd $R/tools && python rocpd_window_stats.py "$(db /tmp/kt_train_$mode)" $ms $STEPS) > $OUT/kernel_stats_train_graph_$mode.csv
  [ $mode == on ] && (cd $R/tools && python rocpd_timeline.py "$(db /tmp/kt_train_$mode)" $ms 0) > $OUT/timeline_train.txt  # idle gaps inside the timed region
done
cp $OUT/kernel_stats_train_graph_on.csv $OUT/kernel_stats_train.csv

# ---- HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, calibrated in the same run on membench
pmc() {  # counter, name, command...
  c=$1; name=$2; shift 2
  lc=$(echo $c | tr 'A-Z' 'a-z')
  rm -rf /tmp/pmc_${c}_$name
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${c}_$name -o pmc -- "$@" > /tmp/pmc_${c}_$name.log 2>&1
  python $R/tools/rocpd_pmc.py "$(db /tmp/pmc_${c}_$name)" > $OUT/pmc_${lc}_$name.csv
}
export EXPO_CHAIN_STREAMS=1  # whole-batch launches: bytes per launch are quoted per whole batch
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr 'A-Z' 'a-z')
  pmc $c chain python $R/bench.py $Q --no-per-kernel --steps 3 --warmup 1
  EXPO_CHAIN_TILE_MIB=0 pmc $c cold python $R/bench.py --shape $COLD $Q --no-per-kernel --steps 2 --warmup 1  # whole-batch launches
  pmc $c infer_B python $R/bench.py --workload infer --shape B --steps 5 --warmup 2
  pmc $c extra python $R/tools/bench_extra.py
  pmc $c chain_fused python $R/bench.py --workload chain_fused --steps 3 --warmup 1
  pmc $c calibration $R/tools/membench 96 9 2 pol
  pmc $c calibration_512 $R/tools/membench 512 9 2 pol
  cp $OUT/pmc_${lc}_chain.csv $OUT/pmc_${lc}.csv
done
unset EXPO_CHAIN_STREAMS
cat $OUT/pytest_gpu.txt
python $R/tools/show_bench.py $OUT/bench_chain.json
