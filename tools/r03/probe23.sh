#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_hip_agent.py tests/test_hip_filters.py -x -q -m gpu 2>&1 | tail -15 | cut -c1-200
