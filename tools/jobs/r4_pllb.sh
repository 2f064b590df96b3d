#!/bin/bash
# PLL block length with the LDS look-ahead ring (c3).  usage: bash tools/jobs/r4_pllb.sh "13312 14976"
for b in ${1:-13312 14976 16640 19968 24960}; do
PDT_PLL_BLOCK=$b python bench.py --config c3 --steps 5 --warmup 1 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('B=$b', d['ms_per_step'], {k:s[k]['ms'] for k in s if k.startswith('pll') or k=='mix_fir'}, 'fixes', d.get('pll_seam_fixes'))"
done
