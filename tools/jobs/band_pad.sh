#!/bin/bash
# Gardner candidate band: pad around the scouts' end points in samples (PDT_BAND_PAD; default 1/8)
for cfg in c2 c3; do for p in 0.125 0.0625 0.03125 0.015625; do
  export PDT_BAND_PAD=$p
  python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$cfg pad $p', d['ms_per_step'], 'table', s['gardner_table']['ms'], 'chain', s['gardner_chain']['ms'], 'walked', d['gardner_walked'], 'cand', d['gardner_candidates'])"
done; done
