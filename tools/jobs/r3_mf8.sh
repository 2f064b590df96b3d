#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_kernel or mix_fir or 250ksps or spanning" 2>&1 | tail -3
for w in 8 4; do
PDT_MF_WAVES=$w timeout 600 python bench.py --config c3 --steps 6 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('waves $w', d['ms_per_step'], 'mix_fir', s['mix_fir']['ms'])"
done
