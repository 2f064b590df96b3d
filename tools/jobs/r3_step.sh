#!/bin/bash
# the 16-operation PLL step: parity subset, then the bench lines it moves
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for c in c3 c2 weak aos argos; do
  python bench.py --config $c --steps 8 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$c', d['ms_per_step'], {k:round(v['ms'],3) for k,v in s.items() if k.startswith('pll') or k in ('mix_fir','agc_block')}, 'fixes', d.get('pll_seam_fixes'))"
done
