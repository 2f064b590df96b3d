#!/bin/bash
# c3: PLL warm-up length against seam repairs
export TMPDIR=/tmp
for ws in 1.0 0.85 0.7 0.55; do
PDT_PLL_WARM_SCALE=$ws timeout 900 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('warm x$ws', d['ms_per_step'], 'phase', s['pll_phase']['ms'], 'head', s['pll_head']['ms'], 'fix', s['pll_fix']['ms'], 'fixes', d.get('pll_seam_fixes'))"
done
