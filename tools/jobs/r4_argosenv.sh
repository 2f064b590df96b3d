for e in "X=1" "PDT_PLL_NOSHORT=1" "PDT_PLL_NOCKPT=1" "PDT_PLL_NOSHORT=1 PDT_PLL_NOCKPT=1"; do
env $e python bench.py --config argos --steps 8 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$e', d['ms_per_step'], {k:s[k]['ms'] for k in s if k.startswith('pll')}, d.get('pll_seam_fixes'))"
done
