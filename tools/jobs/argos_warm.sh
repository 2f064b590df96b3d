#!/bin/bash
# ARGOS: AGC warm-up in gain time constants (PDT_AGC_K, default 34 for double) and PLL / AGC warm-up caps in seconds
run() { python bench.py --config argos --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$1', d['ms_per_step'], 'phase', s['pll_phase']['ms'], 'agc', s['agc_block']['ms'], s['agc_fix']['ms'], 'fixes pll', d['pll_seam_fixes'], 'agc', d['agc_seam_fixes'])"; }
run default
for k in 24 16 12; do export PDT_AGC_K=$k; run "agc_k $k"; done; unset PDT_AGC_K
for w in 1.5 1.0 0.7; do export PDT_AGC_WARM_S=$w; run "agc_warm_s $w"; done; unset PDT_AGC_WARM_S
for w in 1.5 1.0 0.7; do export PDT_PLL_WARM_S=$w; run "pll_warm_s $w"; done; unset PDT_PLL_WARM_S
