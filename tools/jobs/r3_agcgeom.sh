#!/bin/bash
# AGC: smaller look-ahead rings (more walker wavefronts per CU) x shorter blocks, c3
export TMPDIR=/tmp
for lib in "" variants/libpdt_agcpf32.so variants/libpdt_agcpf24.so; do
for tpb in 17 8 5 3 2; do
PDT_LIBPDT_PATH=${lib:+$PWD/$lib} PDT_AGC_TPB=$tpb timeout 600 python bench.py --config c3 --steps 4 --warmup 1 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('lib ${lib:-default} tpb $tpb', d['ms_per_step'], 'agc', s['agc_block']['ms'], 'fix', s['agc_fix']['ms'], d.get('agc_seam_fixes'))"
done
done
