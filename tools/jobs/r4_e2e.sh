#!/bin/bash
# in-process e2e of the c3 line under different overlap settings / CPU bindings.  usage: bash tools/jobs/r4_e2e.sh
run() { # label, env...
  label=$1; shift
  env "$@" PDT_DEBUG_OVERLAP=1 python bench.py --config c3 --steps 2 --warmup 1 --e2e-only 2> /tmp/e2e_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print('$label', 'e2e', e['ms'], e['runs_ms'], e['split_ms'], 'gpu', e['gpu_ms'], d['parity'].get('e2e_text_equals_resident_full_size'))"
  grep "^segment" /tmp/e2e_err.txt | tail -${SEGLINES:-4}
}
run default X=1
run no_overlap PDT_NO_OVERLAP=1
run seg2 PDT_OVERLAP_SEGMENTS=2
run seg3 PDT_OVERLAP_SEGMENTS=3
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist)
echo "node0 $N0 node1 $N1"
label=node0; taskset -c $N0 env PDT_DEBUG_OVERLAP=1 python bench.py --config c3 --steps 2 --warmup 1 --e2e-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print('taskset node0', 'e2e', e['ms'], e['runs_ms'], e['split_ms'])"
taskset -c $N1 env PDT_DEBUG_OVERLAP=1 python bench.py --config c3 --steps 2 --warmup 1 --e2e-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print('taskset node1', 'e2e', e['ms'], e['runs_ms'], e['split_ms'])"
for t in 8 24 32; do PDT_INGEST_THREADS=$t python bench.py --config c3 --steps 2 --warmup 1 --e2e-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print('threads $t', 'e2e', e['ms'], e['runs_ms'], e['split_ms'])"; done
python - <<'PY'
import subprocess,glob
for p in glob.glob('/sys/class/drm/card*/device'):
    try: print(p, open(p+'/numa_node').read().strip(), open(p+'/uevent').read().split('PCI_SLOT_NAME=')[1].split()[0])
    except Exception as e: pass
PY
rocm-smi --showbus 2>/dev/null | head -12
