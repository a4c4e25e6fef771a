#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 600 python tools/r03/probe4.py 2>&1 | grep -v "^aten::miopen_convolution\|^aten::convolution_backward\|amdgpu.ids\|Warning\|_warn" | head -120 | cut -c1-200
