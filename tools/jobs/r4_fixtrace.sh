for b in 0 17472 22464 24960; do
echo "B=$b"; PDT_PLL_BLOCK=$b python bench.py --config c3 --steps 2 --warmup 0 --no-cpu --no-secondary 2>/dev/null | grep "pll fix" | sort | uniq -c | head -8
done
echo weak; python bench.py --config weak --steps 1 --warmup 0 --no-cpu --no-secondary 2>/dev/null | grep "pll fix" | head -40
