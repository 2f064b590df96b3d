#!/bin/bash
# round 3 evidence on the final build: full GPU suite, fuzz (spans / noise leads drawn too), the bench lines of every workload,
# rocprofv3 kernel stats and the HBM counter passes of the c3 line.   usage: bash tools/jobs/r3_evidence.sh <tag> [seed]
TAG=${1:-r3final}; SEED=${2:-301}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 1500 python tests/tools/fuzz.py 300 $SEED > $OUT/fuzz_300_seed$SEED.log 2>&1; tail -2 $OUT/fuzz_300_seed$SEED.log
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default_1gpu.json 2> $OUT/bench_default.err; tail -2 $OUT/bench_default.err
for cfg in c2 argos aos weak; do
  timeout 900 python bench.py --config $cfg --steps 10 --warmup 3 --no-secondary > $OUT/bench_${cfg}_1gpu.json 2> $OUT/bench_$cfg.err; echo "$cfg rc=$?"
done
timeout 600 python bench.py --config c2 --steps 6 --warmup 2 --captures 8 --no-cpu --no-secondary > $OUT/bench_c2_batch8_1gpu.json 2>> $OUT/bench_c2.err
cd /tmp
for cfg in c3 c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o s -- python $R/bench.py --config $cfg --steps 4 --warmup 1 --no-cpu --no-secondary > $OUT/stats_$cfg.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$cfg -o f -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$cfg -o w -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
  cp $(ls $OUT/stats_$cfg/*kernel_stats.csv | head -1) $OUT/rocprofv3_kernel_stats_bench_$cfg.csv
  (cd $R && python tools/pmc_traffic.py $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/pmc_hbm_traffic_bench_$cfg.json > $OUT/pmc_$cfg.txt 2>&1)
  rm -rf $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/stats_$cfg
done
cd $R
python - <<PY
import json
for f in ("bench_default_1gpu", "bench_c2_1gpu", "bench_argos_1gpu", "bench_aos_1gpu", "bench_weak_1gpu", "bench_c2_batch8_1gpu"):
    try:
        d = json.loads(open("$OUT/" + f + ".json").readline())
        print(f, d["value"], "Msps", d["ms_per_step"], "ms; roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"),
              "e2e", d.get("e2e", {}).get("ms"), "cli", d.get("e2e_cli", {}).get("seconds"), "cpu", d.get("cpu_baseline", {}).get("value"), "parity", d.get("parity"))
        print("   ", {k: v["ms"] for k, v in d.get("stages", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
head -20 $OUT/pmc_c3.txt
