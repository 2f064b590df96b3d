#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 300 python tools/probes/dma_ring_beside_kernels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/dma_ring_beside_kernels.txt
