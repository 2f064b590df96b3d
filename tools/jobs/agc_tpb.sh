#!/bin/bash
# AGC block length in FIR tiles (PDT_AGC_TPB; default = 1/16 s of output): lanes per capture vs walkers resident (two per CU: 64 KiB LDS ring each)
for cfg in c3 c2; do for t in 0 12 16 20 28 40; do
  if [ $t = 0 ]; then unset PDT_AGC_TPB; else export PDT_AGC_TPB=$t; fi
  [ $cfg = c2 ] && [ $t -gt 16 ] && continue
  python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$cfg tpb $t', d['ms_per_step'], 'agc', s['agc_block']['ms'], s['agc_fix']['ms'], 'fixes', d['agc_seam_fixes'])"
done; done
for t in 0 3 4 6; do
  if [ $t = 0 ]; then unset PDT_AGC_TPB; else export PDT_AGC_TPB=$t; fi
  python bench.py --config c2 --steps 6 --warmup 2 --captures 8 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('batch8 tpb $t', d['ms_per_step'], 'agc', s['agc_block']['ms'], s['agc_fix']['ms'])"
done
