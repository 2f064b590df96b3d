#!/bin/bash
# round 5: the stream / overlap tests, then the c3 e2e leg with the segments' timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_stream.py -q 2>&1 | tail -80 > gpurun_out/r5/first_stream_tests.log
tail -40 gpurun_out/r5/first_stream_tests.log
PDT_DEBUG_OVERLAP=1 timeout 900 python bench.py --steps 5 --warmup 2 --e2e-only --no-secondary > gpurun_out/r5/first_e2e.json 2> gpurun_out/r5/first_e2e.err
tail -n 75 gpurun_out/r5/first_e2e.err | cut -c1-160
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5/first_e2e.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d.get('e2e'), d.get('parity'))
print({k: v['ms'] for k, v in d['stages'].items()})
PY
timeout 900 python tests/tools/fuzz_segments.py 12 501 2>&1 | tail -16 | tee gpurun_out/r5/first_fuzz_segments.log
