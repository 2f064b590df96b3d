#!/bin/bash
# round 6, first GPU job: the suite on the new build (ABI 4: bounded window, demodMulti passes), then the workloads the round-5 verdict
# asked for -- a pass-shaped capture, the interpolating filter at scale (interp 8 and 3), batched ARGOS -- each with its stages, and
# rocprofv3 kernel statistics of the three POES ones.      usage: bash tools/jobs/r6_first.sh [tag]
TAG=${1:-r6a}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
for cfg in pass i8 c2h; do
  timeout 900 python bench.py --config $cfg --steps 3 --warmup 1 --no-secondary > $OUT/bench_${cfg}_1gpu.json 2> $OUT/bench_$cfg.err; echo "$cfg rc=$?"; tail -2 $OUT/bench_$cfg.err
done
for nc in 32 64; do
  timeout 900 python bench.py --config argos --captures $nc --steps 5 --warmup 2 --no-secondary > $OUT/bench_argos_batch${nc}_1gpu.json 2> $OUT/bench_argos_batch$nc.err; echo "argos x$nc rc=$?"; tail -2 $OUT/bench_argos_batch$nc.err
done
( time timeout 1500 python bench.py --gpus 1 ) > $OUT/bench_default_1gpu.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
cd /tmp
for cfg in i8 c2h pass; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o s -- python $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --no-secondary > $OUT/stats_$cfg.log 2>&1
  cp $(ls $OUT/stats_$cfg/*kernel_stats.csv $OUT/stats_$cfg/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_bench_$cfg.csv 2>/dev/null
  rm -rf $OUT/stats_$cfg
done
cd $R
python - <<PY
import json
for f in ("bench_pass_1gpu", "bench_i8_1gpu", "bench_c2h_1gpu", "bench_argos_batch32_1gpu", "bench_argos_batch64_1gpu", "bench_default_1gpu"):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json") if l.startswith("{")][-1])
        print(f, d["value"], "Msps", d["ms_per_step"], "ms; roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["bound"],
              "e2e", d.get("value_e2e"), d.get("ms_e2e"), "in-process", d.get("e2e", {}).get("ms"), "cpu", d.get("cpu_baseline", {}).get("value"), "parity", d.get("parity"))
        print("   ", {k: v["ms"] for k, v in d.get("stages", {}).items()})
        if "e2e_multi" in d: print("    e2e_multi", {k: d["e2e_multi"].get(k) for k in ("ms", "passes_ms", "until_last_gpu_ms", "gather_ms", "write_ms", "per_gpu", "error")})
    except Exception as e:
        print(f, "failed", e)
PY
