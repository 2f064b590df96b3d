#!/bin/bash
# round 3: mix + FIR in one kernel -- parity tests, then the c3 step with its stage times (fused / unfused)
export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_kernel or mix_fir or spanning or 250ksps or clip_all or alternative or cli" ) > gpurun_out/r3c/pytest_fuse.log 2>&1; tail -15 gpurun_out/r3c/pytest_fuse.log
for v in fused unfused; do
if [ $v = unfused ]; then export PDT_MIX_UNFUSED=1; fi
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-secondary 2> gpurun_out/r3c/bench_$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, 'fir_pll', d['fir_pll_stage'])" || tail -5 gpurun_out/r3c/bench_$v.err
done
