#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_hip_fused_bwd.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_hip_evaluate.py -x -q -k "tf_checkpoint or histogram" 2>&1 | tail -40 | cut -c1-250
