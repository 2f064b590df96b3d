#!/bin/bash
export TMPDIR=/tmp
for ht in 30 36 42 50; do
PDT_HEAD_TAUS=$ht timeout 600 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('head_taus $ht', d['ms_per_step'], 'phase', s['pll_phase']['ms'], 'acq', s['pll_acquire']['ms'], 'head', s['pll_head']['ms'], 'fix', s['pll_fix']['ms'], 'fixes', d.get('pll_seam_fixes'))"
done
for cfg in c2 weak; do for ht in 30 42; do
PDT_HEAD_TAUS=$ht timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$cfg head_taus $ht', d['ms_per_step'], 'phase', s['pll_phase']['ms'], 'head', s['pll_head']['ms'], 'fix', s['pll_fix']['ms'], 'fixes', d.get('pll_seam_fixes'))"
done; done
