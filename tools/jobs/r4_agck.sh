#!/bin/bash
# AGC warm-up length (gain time constants) against walker time and seam repairs.  usage: bash tools/jobs/r4_agck.sh "10 11 12" "c3 aos"
for cfg in ${2:-c3 aos weak c2}; do for k in ${1:-10 11 12 13 14 16}; do
PDT_AGC_K=$k python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg K=$k', d['ms_per_step'], 'agc', d['stages']['agc_block']['ms'], 'fix', d['stages']['agc_fix']['ms'], 'agc fixes', d.get('agc_seam_fixes'))"
done; done
