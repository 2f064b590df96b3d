#!/bin/bash
# the phase walkers' block size at c3 (PDT_PLL_BLOCK; 0 = the library's choice), two rounds
for rep in 1 2; do for b in ${BLOCKS:-0 19968 22464 23296 24128 24960 26624 28288}; do
PDT_PLL_BLOCK=$b python bench.py --config c3 --steps 8 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('B=$b', d['ms_per_step'], {k:s[k]['ms'] for k in s if k.startswith('pll') or k=='mix_fir'}, 'fixes', d.get('pll_seam_fixes'))"
done; done
