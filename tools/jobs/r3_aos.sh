#!/bin/bash
# round 3: the pre-lock (aos) and weak-signal (weak) workloads, full lines with parity and CPU legs
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
for cfg in aos weak; do
( time timeout 1500 python bench.py --config $cfg --steps 5 --warmup 1 --no-secondary ) > gpurun_out/r3e/bench_$cfg.json 2> gpurun_out/r3e/bench_$cfg.err; tail -3 gpurun_out/r3e/bench_$cfg.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3e/bench_$cfg.json').readline())
    print('$cfg', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, d.get('parity'), 'cpu', d.get('cpu_baseline',{}).get('value'), 'fixes', d.get('pll_seam_fixes'))
except Exception as e:
    print('$cfg failed', e)
PY
done
