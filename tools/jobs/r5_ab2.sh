#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 900 python tools/e2e_ab.py 24 plain=PDT_NO_OVERLAP:1 s3_noearly=PDT_NO_EARLY_PLL:1 s3_early=PDT_X:0 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r5/ab_early.txt
