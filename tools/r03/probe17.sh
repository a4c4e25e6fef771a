#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_hip_agent.py -q -m gpu -k "fused" 2>&1 | grep -v "^  File\|^$" | tail -60 | cut -c1-250
