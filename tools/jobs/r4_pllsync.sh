#!/bin/bash
# the phase walkers alone (every launch followed by a stream synchronisation) against beside acquisition + head
export TMPDIR=/tmp
for e in "X=1" "PDT_DEBUG_SYNC=1"; do
env $e python bench.py --config c3 --steps 4 --warmup 1 --no-cpu --no-secondary 2>/tmp/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$e', d['ms_per_step'], {k:s[k]['ms'] for k in s if k.startswith('pll')})"
grep -m2 "k_pll_phase\|pll_phase" /tmp/err.log
done
