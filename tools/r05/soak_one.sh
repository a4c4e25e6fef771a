#!/bin/bash
# One test (pytest node id or -k expression) N times in fresh processes; stops at the first failing run and prints it.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
N=${2:-20}
for i in $(seq 1 $N); do
  python -m pytest $R/tests -x -q -m gpu -k "$1" > /tmp/one_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "run $i: rc=$rc"; grep -v "^\s*$" /tmp/one_$i.log | tail -60 | cut -c1-300; exit 1
  fi
done
echo "$N runs of '$1': all passed"
