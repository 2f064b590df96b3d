#!/bin/bash
# does the shader clock under the benched step differ from the one a lone probe kernel sees?  (rocm-smi sampled beside both)
export TMPDIR=/tmp
sample() { for i in $(seq 1 $1); do rocm-smi -c -P 2>/dev/null | grep -i "sclk\|mclk\|power (W)\|Socket" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.4; done; }
echo "== idle"; sample 2
echo "== bench c3, 500 steps"
python bench.py --config c3 --steps 500 --warmup 20 --no-cpu --no-secondary > /tmp/b.json 2>/dev/null &
BP=$!
sleep 14; sample 12
wait $BP
python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('c3', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
echo "== probe"
./tools/probes/pll_mem_probe > /tmp/p.log 2>&1 &
PP=$!
sleep 3; sample 8
wait $PP; grep -i "library" /tmp/p.log | head -3
