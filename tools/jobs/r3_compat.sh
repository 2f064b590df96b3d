#!/bin/bash
# round 3: the link-compatible shim + the fused kernel again + full GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
( time timeout 1200 python -m pytest tests/test_gpu_compat.py -m gpu -x -q ) > gpurun_out/r3d/pytest_compat.log 2>&1; tail -25 gpurun_out/r3d/pytest_compat.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu --no-secondary 2> gpurun_out/r3d/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, 'fir_pll', d['fir_pll_stage'])" || tail -5 gpurun_out/r3d/bench.err
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > gpurun_out/r3d/pytest_all.log 2>&1; tail -8 gpurun_out/r3d/pytest_all.log
