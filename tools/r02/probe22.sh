#!/bin/bash
# dispatch backward: curve launch on the library's side stream (fork/join) vs the serial pair vs HEAD
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p22
mkdir -p $OUT
for rep in 1 2; do
  EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so timeout 200 python tools/bench_extra.py > $OUT/head_$rep.json 2>/dev/null
  EXPO_DISPATCH_FORK=0 timeout 200 python tools/bench_extra.py > $OUT/serial_$rep.json 2>/dev/null
  EXPO_DISPATCH_FORK=1 timeout 200 python tools/bench_extra.py > $OUT/fork_$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02p22/*_?.json')):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])['kernels']
    print(f.split('/')[-1], {k: round(v['ms'] * 1e3, 1) for k, v in d.items() if 'dispatch' in k or 'stats' in k})
  except Exception as e:
    print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_fork
EXPO_DISPATCH_FORK=1 rocprofv3 --kernel-trace -d /tmp/kt_fork -o kt -- python $R/tools/bench_extra.py > /tmp/kt_fork.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/kt_fork/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
v = [t for t in tabs if t == 'kernels'] or [t for t in tabs if 'kernel' in t.lower()]
print(v[:5])
cols = [r[1] for r in c.execute("pragma table_info(%s)" % v[0])]
print(cols)
rows = c.execute("select name, start, end, stream_id, queue_id from %s where name like '%%dispatch_bwd%%' order by start limit 400" % v[0]).fetchall()
print(len(rows))
for r in rows[-12:]:
  print(r[0][:40].split('<')[0], r[1] % 10**9, r[2] - r[1], r[3], r[4])
PY
