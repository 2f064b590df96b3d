#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 900 python tools/e2e_ab.py 4 plain=PDT_NO_OVERLAP:1,PDT_DEBUG_OVERLAP:1 overlapped=PDT_DEBUG_OVERLAP:1 2>&1 | grep -v "amdgpu.ids\|^    " | tail -60
