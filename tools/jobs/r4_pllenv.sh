for e in "X=1" "PDT_NO_EXCL=1" "PDT_ACQUIRE_SIMPLE=1" "PDT_HEAD_TAUS=30"; do
env $e python bench.py --config c3 --steps 6 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$e', d['ms_per_step'], {k:s[k]['ms'] for k in s if k.startswith('pll')})"
done
