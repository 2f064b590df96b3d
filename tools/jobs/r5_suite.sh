#!/bin/bash
# the full GPU suite + the segment fuzz
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > gpurun_out/r5/pytest_gpu.log 2>&1; tail -15 gpurun_out/r5/pytest_gpu.log
timeout 900 python tests/tools/fuzz_segments.py 30 511 > gpurun_out/r5/fuzz_segments_30.log 2>&1; tail -2 gpurun_out/r5/fuzz_segments_30.log
timeout 300 python tools/probes/stage_concurrency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/stage_concurrency.txt
