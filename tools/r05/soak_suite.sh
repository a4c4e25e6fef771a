#!/bin/bash
# The whole gpu suite N times; on a failing run print its tail (pytest's faulthandler dump included) and stop.
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-3}
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  EXPO_RECORD_PARAM_ERR=/tmp/pe_$i.jsonl python -X faulthandler -m pytest $R/tests -x -q -m gpu > /tmp/suite_$i.log 2>&1
  rc=$?
  echo "run $i: rc=$rc $(tail -1 /tmp/suite_$i.log | cut -c1-120)"
  if [ $rc -ne 0 ]; then
    echo "---- failing run $i ----"
    grep -v "^\s*$" /tmp/suite_$i.log | tail -120 | cut -c1-300
    dmesg 2>/dev/null | tail -20
    break
  fi
done
