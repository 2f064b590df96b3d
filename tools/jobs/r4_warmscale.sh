for cfg in c3 weak; do for sc in 1.0 0.85 0.7 0.55; do
PDT_PLL_WARM_SCALE=$sc python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('$cfg scale $sc', d['ms_per_step'], {k:s[k]['ms'] for k in s if k.startswith('pll')}, 'fixes', d.get('pll_seam_fixes'))"
done; done
