#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_stages.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3
for cfg in c3 c2 weak aos argos; do
timeout 600 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$cfg', d['ms_per_step'], 'phase', s['pll_phase']['ms'], 'acq', s['pll_acquire']['ms'], 'head', s['pll_head']['ms'], 'fix', s['pll_fix']['ms'], 'fixes', d.get('pll_seam_fixes'))"
done
