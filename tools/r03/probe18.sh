#!/bin/bash
# r03p18: full gpu suite + default bench line + infer line after the round's kernel changes
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p18
rm -rf $OUT; mkdir -p $OUT
cd $R
( time timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python tools/show_bench.py $OUT/bench.json 2>/dev/null | head -40
python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']
print({k: r[k] for k in ('kernel','achieved','frac','avg_launch_ms','avg_launch_ms_event_pair_per_launch','event_pair_overhead_ms','rocprof_avg_us','chain_frac')})
print(d['cpu_baseline'].get('parity_check'))
print('value', d['value'], 'ms', d['ms_per_step'])
"
for s in B C; do python bench.py --workload infer --shape $s --steps 50 --warmup 10 > $OUT/infer_$s.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/infer_$s.json')); r=d['roofline']; print('$s', d['ms_per_step']*1e3, {k: r[k] for k in ('achieved','peak','frac','avg_launch_ms','issue_floor_ms','hbm_frac_of_8TBps')})"; done
python -c "import __graft_entry__ as g; g.smoke()"
