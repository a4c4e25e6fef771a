#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p14
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/kt_train -o kt -- python $R/bench.py --workload train --no-cpu-baseline --steps 20 --warmup 6 > /tmp/kt_train.log 2>&1
tail -2 /tmp/kt_train.log
python - <<PY
import sqlite3, glob, re
db = glob.glob('/tmp/kt_train/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
# keep only the timed region: the last 20 iterations ~ last 20/26 of the kernel launches by time
rows = c.execute('select name, start, end from kernels order by start').fetchall()
t_end = rows[-1][2]
# find start of timed region: take launches in the last 20*21.8ms window before end minus 0.2 s of teardown
import collections
win0 = t_end - int(20*21.8e6) - int(5e6)
agg = collections.defaultdict(lambda: [0,0])
for n,s,e in rows:
    if s >= win0:
        a = agg[n]; a[0]+=1; a[1]+=e-s
tot = sum(v[1] for v in agg.values())
out = open('$OUT/train_kernels.csv','w')
out.write('"Name","CallsPerIter","TotalMsPerIter","AvgUs","Pct"\n')
for n,(cnt,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:60]:
    out.write('"%s",%.1f,%.3f,%.2f,%.2f\n' % (re.sub(r'\s+',' ',n)[:150], cnt/20, t/20/1e6, t/cnt/1e3, 100*t/tot))
out.write('"TOTAL",%.1f,%.3f,,100\n' % (sum(v[0] for v in agg.values())/20, tot/20/1e6))
PY
head -45 $OUT/train_kernels.csv | cut -c1-200
