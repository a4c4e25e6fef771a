#!/bin/bash
# where does the training iteration (BASELINE config 3) spend its 21 ms?
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_train
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_train -o kt -- python $R/bench.py --workload train --steps 10 --warmup 3 > $OUT/train.log 2>&1
tail -1 $OUT/train.log | cut -c1-400
python $R/tools/rocpd_stats.py "$(find /tmp/kt_train -name '*.db' | head -1)" > $OUT/kernel_stats_train.csv
head -45 $OUT/kernel_stats_train.csv | cut -c1-200
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/root/repo/gpurun_out/r02p25/kernel_stats_train.csv')))
tot = sum(int(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('kernels', len(rows), 'calls', calls, 'total ms', tot / 1e6)
PY
