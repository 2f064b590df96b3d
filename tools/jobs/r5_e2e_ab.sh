#!/bin/bash
# e2e A/B on ONE box: the capture ingested first (PDT_NO_OVERLAP) against the overlapped segments
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --e2e-only --no-secondary > gpurun_out/r5/ab_$name.json 2> gpurun_out/r5/ab_$name.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r5/ab_$name.json') if l.startswith('{')][-1])
print('$name', d['e2e']['ms'], d['e2e']['runs_ms'], d['parity'].get('e2e_text_equals_resident_full_size'))
PY
}
run plain PDT_NO_OVERLAP=1
run overlap PDT_DEBUG_OVERLAP=1
grep "^segment" gpurun_out/r5/ab_overlap.err | tail -3
run overlap_2stream PDT_INGEST_STREAMS=2
run overlap_3stream PDT_INGEST_STREAMS=3
run overlap_t12 PDT_INGEST_THREADS=12
run overlap_split PDT_OVERLAP_SPLIT=0.67,0.205,0.125
run plain2 PDT_NO_OVERLAP=1
run overlap2 PDT_DEBUG_OVERLAP=1
grep "^segment" gpurun_out/r5/ab_overlap2.err | tail -3
timeout 300 python tools/probes/dma_ring_beside_kernels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/dma_ring_beside_kernels.txt
