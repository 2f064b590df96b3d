#!/bin/bash
# the early theta / phase kernels of the overlapped ingest: parity (stream tests, segment fuzz), then the interleaved e2e A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stream.py -q 2>&1 | tail -12
timeout 900 python tests/tools/fuzz_segments.py 24 521 2>&1 | tail -4
timeout 900 python tools/e2e_ab.py 10 plain=PDT_NO_OVERLAP:1 s3_noearly=PDT_NO_EARLY_PLL:1 s3=PDT_DEBUG_OVERLAP:0 s4b=PDT_OVERLAP_SPLIT:0.42/0.27/0.18/0.13 s5=PDT_OVERLAP_SPLIT:0.40/0.22/0.16/0.12/0.10 2>&1 | grep -v "amdgpu.ids\|^segment\|^    \|ingest_capture" | tee gpurun_out/r5/ab_early.txt
