#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > gpurun_out/r3g/pytest_all.log 2>&1; tail -6 gpurun_out/r3g/pytest_all.log
for cfg in c3 c2 weak argos; do
timeout 900 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$cfg', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in s.items()}, 'fixes', d.get('pll_seam_fixes'), d.get('agc_seam_fixes'))"
done
PDT_AGC_ONEPASS=1 timeout 900 python bench.py --config c3 --steps 8 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('c3 onepass', d['ms_per_step'], 'agc', s['agc_block']['ms'])"
