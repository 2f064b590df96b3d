#!/bin/bash
# e2e of the c3 line, the round's PLL additions switched off and on, alternating on the same box
for rep in 1 2; do for e in "X=1" "PDT_PLL_NOCKPT=1 PDT_PLL_NOCONSENSUS=1 PDT_PLL_NOSHORT=1"; do
env $e python bench.py --config c3 --steps 2 --warmup 1 --e2e-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print('$e', 'e2e', e['ms'], e['runs_ms'], e['split_ms'], 'gpu', e['gpu_ms'])"
done; done
