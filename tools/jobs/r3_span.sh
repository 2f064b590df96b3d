#!/bin/bash
# round 3: table rows that span several chunks -- parity tests, then the c3 step with its stage times
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spanning or 250ksps or clip_all or synthetic_rates or alternative" ) > gpurun_out/r3b/pytest_span.log 2>&1; tail -15 gpurun_out/r3b/pytest_span.log
for sp in 1 8 16 32; do
PDT_GSPAN=$sp timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-secondary 2> gpurun_out/r3b/bench_span$sp.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('span $sp', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, 'walked', d.get('gardner_walked'), 'cand', d.get('gardner_candidates'))" || tail -5 gpurun_out/r3b/bench_span$sp.err
done
