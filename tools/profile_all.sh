#!/bin/bash
# Regenerates every measured artefact under profiles/ in ONE gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/profile_all.sh'
# then, back in the container:  tools/collect_profiles.sh  (copies gpurun_out/final/* to profiles/r01_final_* and
# rebuilds profiles/traffic.json with tools/make_traffic.py)
# rocprofv3 passes: --kernel-trace alone (durations) and one --pmc counter per pass (never combined
# with sys/runtime tracing); the rocpd databases stay on the GPU box, only CSV summaries come back.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }

python -m pytest $R/tests -x -q -m gpu 2>&1 | tail -2 > $OUT/pytest_gpu.txt
python $R/bench.py > $OUT/bench_chain.json 2> $OUT/bench_chain.err
python $R/bench.py --shape A --no-cpu-baseline > $OUT/bench_chain_A.json 2>/dev/null
python $R/bench.py --dtype f32 --no-cpu-baseline > $OUT/bench_chain_f32.json 2>/dev/null
python $R/bench.py --shape B --no-cpu-baseline > $OUT/bench_chain_B.json 2>/dev/null
python $R/bench.py --workload infer --shape B --no-cpu-baseline > $OUT/bench_infer_B.json 2>/dev/null
python $R/bench.py --workload train --no-cpu-baseline > $OUT/bench_train.json 2>/dev/null
python $R/tools/bench_extra.py > $OUT/bench_extra.json 2>/dev/null
$R/tools/membench > $OUT/membench.txt 2>&1

rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline > /tmp/kt.log 2>&1
python $R/tools/rocpd_stats.py "$(db /tmp/kt)" > $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o pmc -- python $R/bench.py --no-cpu-baseline --no-per-kernel --steps 3 --warmup 1 > /tmp/pmc_$c.log 2>&1
  python $R/tools/rocpd_pmc.py "$(db /tmp/pmc_$c)" > $OUT/pmc_$lc.csv
  # calibration of the counter on kernels with a known byte count and the SAME access pattern and
  # cache policy (membench cpol / rpol move exactly 96 MiB per stream)
  rocprofv3 --pmc $c --kernel-trace -d /tmp/cal_$c -o cal -- $R/tools/membench 96 9 2 pol > /tmp/cal_$c.log 2>&1
  python $R/tools/rocpd_pmc.py "$(db /tmp/cal_$c)" > $OUT/pmc_${lc}_calibration.csv
done
cat $OUT/pytest_gpu.txt
python $R/tools/show_bench.py $OUT/bench_chain.json
