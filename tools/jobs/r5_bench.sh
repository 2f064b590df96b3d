#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( time timeout 1500 python bench.py --gpus 1 ) > gpurun_out/r5/bench_default_1gpu.json 2> gpurun_out/r5/bench_default.err; tail -3 gpurun_out/r5/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5/bench_default_1gpu.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], 'e2e', d['e2e']['ms'], d['e2e']['runs_ms'], 'first', d['e2e']['first_run_ms'])
print('cli', d['e2e_cli']); print('cli -P', d.get('e2e_cli_noprogress'))
print('cpu', d['cpu_baseline']['value'], d.get('cpu_baseline_8proc', {}).get('value'), 'parity', d['parity'])
print({k: v['ms'] for k, v in d['stages'].items()})
print('secondary', {k: (v.get('value'), v.get('ms_per_step')) for k, v in d.get('secondary', {}).items()})
PY
