#!/bin/bash
# e2e A/B on ONE box: the capture ingested first (PDT_NO_OVERLAP) against the overlapped segments, splits
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
run s4b PDT_DEBUG_OVERLAP=1 PDT_OVERLAP_SPLIT=0.42,0.27,0.18,0.13
grep "^segment" gpurun_out/r5/ab_s4b.err | tail -4
run s3b PDT_OVERLAP_SPLIT=0.55,0.28,0.17
run s4 PDT_OVERLAP_SPLIT=0.48,0.25,0.15,0.12
run s4c PDT_OVERLAP_SPLIT=0.38,0.27,0.20,0.15
run s4b_t12 PDT_OVERLAP_SPLIT=0.42,0.27,0.18,0.13 PDT_INGEST_THREADS=12
run s4b_t16 PDT_OVERLAP_SPLIT=0.42,0.27,0.18,0.13 PDT_INGEST_THREADS=16
run plain_t12 PDT_NO_OVERLAP=1 PDT_INGEST_THREADS=12
run plain2 PDT_NO_OVERLAP=1
run s4b_again PDT_DEBUG_OVERLAP=1 PDT_OVERLAP_SPLIT=0.42,0.27,0.18,0.13
grep "^segment" gpurun_out/r5/ab_s4b_again.err | tail -4
