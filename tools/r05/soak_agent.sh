#!/bin/bash
# Repeats the gpu tests of one file until one run dies; prints the tail of the failing run (faulthandler dump included).
R=${GRAFT_REPO_ROOT:-$PWD}
FILE=${1:-tests/test_hip_agent.py}
N=${2:-6}
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  python -m pytest $R/$FILE -x -q -m gpu > /tmp/soak_$i.log 2>&1
  rc=$?
  echo "run $i: rc=$rc $(tail -1 /tmp/soak_$i.log | cut -c1-120)"
  if [ $rc -ne 0 ]; then
    echo "---- failing run $i ----"
    grep -v "^\s*$" /tmp/soak_$i.log | tail -60 | cut -c1-300
    break
  fi
done
