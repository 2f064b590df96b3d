#!/bin/bash
# c3: PLL block size sweep (PDT_PLL_BLOCK, samples); default = capture / (240 CUs x 256 lanes)
for b in 0 20032 24000 32000 48000 64000; do
  if [ $b = 0 ]; then unset PDT_PLL_BLOCK; else export PDT_PLL_BLOCK=$b; fi
  echo "block $b"; python bench.py --config c3 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if k.startswith('pll')}, 'fixes', d['pll_seam_fixes'])"
done
