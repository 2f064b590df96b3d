#!/bin/bash
# c3 step against the PLL block length (blocks of a multiple of 832 samples keep k_mix_fir): fewer, longer blocks re-read less
export TMPDIR=/tmp
for b in ${PLLB_LIST:-14976 19968 24960 29952 39936 49920}; do
  PDT_PLL_BLOCK=$b python bench.py --config c3 --steps 6 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('B', $b, 'c3', d['ms_per_step'], 'phase', s['pll_phase']['ms'], 'acq', s['pll_acquire']['ms'], 'head', s['pll_head']['ms'], 'fix', s['pll_fix']['ms'], 'mixfir', s['mix_fir']['ms'], 'fixes', d.get('pll_seam_fixes'))"
done
