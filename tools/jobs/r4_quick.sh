#!/bin/bash
# the four bench lines with their stage times + the parity files (usage: bash tools/jobs/r4_quick.sh [pytest files...])
export TMPDIR=/tmp
for cfg in ${CFGS:-c3 c2 weak argos}; do python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, 'fixes', d.get('pll_seam_fixes'), d.get('agc_seam_fixes'))"; done
if [ -n "$BATCH" ]; then python bench.py --config c2 --steps 6 --warmup 2 --captures 8 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch8', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"; fi
[ $# -gt 0 ] && timeout 1500 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -4
