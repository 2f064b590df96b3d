#!/bin/bash
# round 6, second GPU job: the suite on the split build with the round's kernel changes (acquisition loop in four-sample blocks, tail
# pass beside the acquisition, sampler rows sized by the chunk's symbols, batch geometry of the ARGOS walkers), the workloads they aim at.
TAG=${1:-r6b}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
for cfg in pass aos i8 c2h weak; do
  timeout 900 python bench.py --config $cfg --steps 3 --warmup 1 --no-secondary > $OUT/bench_${cfg}_1gpu.json 2> $OUT/bench_$cfg.err; echo "$cfg rc=$?"; tail -2 $OUT/bench_$cfg.err | cut -c1-600
done
PDT_PLL_NOTAIL=1 timeout 900 python bench.py --config pass --steps 2 --warmup 1 --no-secondary --no-cpu > $OUT/bench_pass_notail_1gpu.json 2> $OUT/bench_pass_notail.err
for nc in 1 32 64; do
  timeout 900 python bench.py --config argos --captures $nc --steps 5 --warmup 2 --no-secondary > $OUT/bench_argos_batch${nc}_1gpu.json 2> $OUT/bench_argos_batch$nc.err; echo "argos x$nc rc=$?"; tail -2 $OUT/bench_argos_batch$nc.err | cut -c1-600
done
timeout 900 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu > $OUT/bench_c3_resident_1gpu.json 2> $OUT/bench_c3.err
python - <<PY
import json
for f in ("bench_pass_1gpu", "bench_pass_notail_1gpu", "bench_aos_1gpu", "bench_weak_1gpu", "bench_i8_1gpu", "bench_c2h_1gpu", "bench_argos_batch1_1gpu", "bench_argos_batch32_1gpu", "bench_argos_batch64_1gpu", "bench_c3_resident_1gpu"):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json") if l.startswith("{")][-1])
        par = d.get("parity", {})
        bad = [k for k, v in par.items() if v is False or (isinstance(v, list) and v and isinstance(v[0], bool) and not all(v))]
        print(f, d["value"], "Msps", d["ms_per_step"], "ms; e2e", d.get("value_e2e"), d.get("ms_e2e"), "cpu", d.get("cpu_baseline", {}).get("value"), "parity bad:", bad, "fixes", d.get("pll_seam_fixes"), d.get("agc_seam_fixes"))
        print("   ", {k: v["ms"] for k, v in d.get("stages", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
