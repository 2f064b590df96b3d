#!/bin/bash
export TMPDIR=/tmp
for cfg in c3 c2 weak argos; do
timeout 900 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$cfg', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in s.items()}, 'fixes', d.get('pll_seam_fixes'))"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "weak or noise or snr or block_geometry" 2>&1 | tail -3
