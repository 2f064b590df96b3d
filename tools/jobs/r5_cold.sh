#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
( ./tools/probes/cold_path_probe; echo; echo "second process:"; ./tools/probes/cold_path_probe ) 2>&1 | tee gpurun_out/r5/cold_path_probe.txt
