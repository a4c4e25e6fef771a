#!/bin/bash
# Copies what tools/profile_all.sh wrote under gpurun_out/final/ into profiles/ (tracked) and rebuilds
# profiles/traffic.json; tests/test_profiles_consistency.py then checks the set is consistent.
set -euo pipefail
cd "$(dirname "$0")/.."
SRC=gpurun_out/final
TAG=${1:-r06_final}
for f in bench_chain bench_chain_1stream bench_chain_A bench_chain_B bench_chain_f32 bench_chain_cold bench_chain_cold_untiled bench_infer_B bench_infer_C bench_chain_fused bench_chain_fused_B bench_chain_fused_f32_B bench_train bench_train_eager bench_extra; do
  [ -s $SRC/$f.json ] && cp $SRC/$f.json profiles/${TAG}_$f.json
done
for f in $SRC/kernel_stats*.csv $SRC/pmc_*.csv $SRC/membench*.txt $SRC/conv_bench.txt $SRC/conv_sweep.txt $SRC/step_bench.txt $SRC/step_trace.txt $SRC/iter_probe.txt $SRC/timeline_train.txt $SRC/pytest_gpu.txt; do
  [ -s $f ] || continue
  cp $f profiles/${TAG}_$(basename $f)
done
# bench.py quotes, beside its own figure, rocprofv3's duration of the dominant kernel from the table committed WHEN IT
# RAN (the previous collection); point the collected lines at the table of their own gpurun call instead
python - "$TAG" <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
tag = sys.argv[1]
for name in ('bench_chain', 'bench_chain_1stream', 'bench_chain_A', 'bench_chain_B'):
  path = 'profiles/%s_%s.json' % (tag, name)
  try:
    d = json.load(open(path))
  except OSError:
    continue
  legs = d.get('config', {}).get('legs')
  if legs is not None:  # (the same re-pointing for the launch count the line quotes from the committed training table)
    legs['train_launches_per_iteration'], legs['train_launches_source'] = bench.committed_train_launches()
    json.dump(d, open(path, 'w'))
  r = d.get('roofline', {})
  if 'kernel' in r and 'rocprof_avg_us' in r:
    c = d['config']
    shape = (c['batch_per_gpu'], c['height'], c['width'], 3)
    r['rocprof_avg_us'] = bench.rocprof_avg_us(r['kernel'], shape, d['dtype'], prefix=tag.split('_')[0])
    json.dump(d, open(path, 'w'))
PY
[ -s $SRC/param_grad_errors.jsonl ] && python tools/r04/param_err_table.py $SRC/param_grad_errors.jsonl > profiles/${TAG}_param_grad_errors.md
python tools/make_traffic.py $SRC profiles/traffic.json 64x512x512x3:f16 > /dev/null
python tools/make_traffic.py $SRC profiles/traffic.json 256x512x512x3:f16 cold > /dev/null
python tools/kernel_table.py $SRC > profiles/${TAG}_kernel_table.md
python -m pytest tests/test_profiles_consistency.py -q
