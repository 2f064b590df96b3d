#!/bin/bash
# Gardner scouts: symbols of the previous chunk they run over (PDT_SCOUT_SYMS; 455 = the 4096 samples of round 1 at 50 ksps)
for cfg in c2 c3; do for n in 455 273 200 140 100; do
  export PDT_SCOUT_SYMS=$n
  python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$cfg syms $n', d['ms_per_step'], 'table', s['gardner_table']['ms'], 'chain', s['gardner_chain']['ms'], 'walked', d['gardner_walked'], 'cand', d['gardner_candidates'])"
done; done
for n in 455 273 200 140; do export PDT_SCOUT_SYMS=$n
python bench.py --config c2 --steps 6 --warmup 2 --captures 8 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('batch8 syms $n', d['ms_per_step'], 'table', s['gardner_table']['ms'])"
done
