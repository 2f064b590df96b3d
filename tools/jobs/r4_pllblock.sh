for b in 0 16640 17472 18304 21632 22464 24960; do
PDT_PLL_BLOCK=$b python bench.py --config c3 --steps 8 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']; print('B=$b', d['ms_per_step'], {k:s[k]['ms'] for k in s if k.startswith('pll') or k=='mix_fir'}, 'fixes', d.get('pll_seam_fixes'))"
done
