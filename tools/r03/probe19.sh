#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_x
rocprofv3 --kernel-trace -d /tmp/kt_x -o kt -- python $R/tools/bench_extra.py --shape A > /dev/null 2>&1
db=$(find /tmp/kt_x -name "*.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$db")
print([r[1] for r in c.execute("PRAGMA table_info(kernels)")])
print(c.execute("select * from kernels limit 1").fetchall())
PY
