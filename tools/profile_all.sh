#!/bin/bash
# Regenerates every measured artefact under profiles/ in ONE gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/profile_all.sh'
# then, back in the container:  tools/collect_profiles.sh r02_final
# rocprofv3 passes: --kernel-trace alone (durations) and one --pmc counter per pass (never combined
# with sys/runtime tracing); the rocpd databases stay on the GPU box, only CSV summaries come back.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
COLD=256,512,512
Q="--no-cpu-baseline --cold-shape none"

python -m pytest $R/tests -x -q -m gpu 2>&1 | tail -2 > $OUT/pytest_gpu.txt
python $R/bench.py > $OUT/bench_chain.json 2> $OUT/bench_chain.err
python $R/bench.py --shape A $Q > $OUT/bench_chain_A.json 2>/dev/null
python $R/bench.py --shape B $Q > $OUT/bench_chain_B.json 2>/dev/null
python $R/bench.py --dtype f32 $Q > $OUT/bench_chain_f32.json 2>/dev/null
python $R/bench.py --shape $COLD $Q > $OUT/bench_chain_cold.json 2>/dev/null
python $R/bench.py --workload infer --shape B > $OUT/bench_infer_B.json 2>/dev/null
python $R/bench.py --workload infer --shape C > $OUT/bench_infer_C.json 2>/dev/null
python $R/bench.py --workload train --no-cpu-baseline > $OUT/bench_train.json 2>/dev/null
python $R/tools/bench_extra.py > $OUT/bench_extra.json 2>/dev/null
for sz in 96 512 1024; do
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
kt chain python $R/bench.py $Q
kt cold python $R/bench.py --shape $COLD $Q
kt chain_B python $R/bench.py --shape B $Q
kt infer_B python $R/bench.py --workload infer --shape B
kt infer_C python $R/bench.py --workload infer --shape C
kt extra python $R/tools/bench_extra.py
cp $OUT/kernel_stats_chain.csv $OUT/kernel_stats.csv

# ---- HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, calibrated in the same run on membench
pmc() {  # counter, name, command...
  c=$1; name=$2; shift 2
  lc=$(echo $c | tr 'A-Z' 'a-z')
  rm -rf /tmp/pmc_${c}_$name
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${c}_$name -o pmc -- "$@" > /tmp/pmc_${c}_$name.log 2>&1
  python $R/tools/rocpd_pmc.py "$(db /tmp/pmc_${c}_$name)" > $OUT/pmc_${lc}_$name.csv
}
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr 'A-Z' 'a-z')
  pmc $c chain python $R/bench.py $Q --no-per-kernel --steps 3 --warmup 1
  pmc $c cold python $R/bench.py --shape $COLD $Q --no-per-kernel --steps 2 --warmup 1
  pmc $c infer_B python $R/bench.py --workload infer --shape B --steps 5 --warmup 2
  pmc $c extra python $R/tools/bench_extra.py
  pmc $c calibration $R/tools/membench 96 9 2 pol
  pmc $c calibration_512 $R/tools/membench 512 9 2 pol
  cp $OUT/pmc_${lc}_chain.csv $OUT/pmc_${lc}.csv
done
cat $OUT/pytest_gpu.txt
python $R/tools/show_bench.py $OUT/bench_chain.json
