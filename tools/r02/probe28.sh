#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p28
mkdir -p $OUT
T0=$(date +%s); python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "Elapsed $(( $(date +%s) - T0 )) s" >> $OUT/bench.err
grep "cpu_baseline\|Elapsed" $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02p28/bench.json').read().strip().splitlines()[-1])
c = d['cpu_baseline']
print(d['ms_per_step'], d['value'], d['roofline']['frac'])
print(json.dumps({k: c[k] for k in ('value', 'cores', 'c_port_sweep', 'seconds')}))
print({k: (v['best_threads'], round(v['fwd_bwd_Mpixels_per_s'], 2)) for k, v in c['op_by_op'].items()})
PY
python -c "import __graft_entry__ as g; g.smoke()"
