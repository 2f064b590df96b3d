#!/bin/bash
# quick A/B: bench line of the current build (optionally PDT_LIBPDT_PATH variants). usage: bash tools/jobs/one.sh [extra bench args]
python bench.py --steps 10 --warmup 3 --no-cpu "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if k.startswith('pll')}, 'fixes', d['pll_seam_fixes'])"
