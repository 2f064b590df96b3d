#!/bin/bash
# FIR: persistent workgroups per CU (PDT_FIR_WG_PER_CU; default = what fits)
for cfg in c2 c3; do for w in 4 8 16 32 64 4096; do
  if [ $w = 0 ]; then unset PDT_FIR_WG_PER_CU; else export PDT_FIR_WG_PER_CU=$w; fi
  python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg wg/cu $w', d['ms_per_step'], 'fir', d['stages']['fir']['ms'])"
done; done
