#!/bin/bash
# parity tests of the main file + the three bench lines (stages).  usage: bash tools/jobs/quick.sh [pytest -k expr]
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -x -q ${1:+-k "$1"} 2>&1 | tail -4
for cfg in c2 c3; do
python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, 'fixes', d['pll_seam_fixes'])"
done
python bench.py --config c2 --steps 6 --warmup 2 --captures 8 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch8', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
