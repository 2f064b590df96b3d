#!/bin/bash
# round 3: span kernels (keys / walk / join), per-chunk lane emission, quality on the side stream, AGC warm-up cap
export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_quality.py tests/test_gpu_batch.py -m gpu -x -q ) > gpurun_out/r3f/pytest.log 2>&1; tail -12 gpurun_out/r3f/pytest.log
for cfg in c3 aos weak; do
timeout 900 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu --no-secondary 2> gpurun_out/r3f/bench_$cfg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, 'walked', d.get('gardner_walked'), 'cand', d.get('gardner_candidates'), 'fixes', d.get('pll_seam_fixes'), d.get('agc_seam_fixes'))" || tail -5 gpurun_out/r3f/bench_$cfg.err
done
