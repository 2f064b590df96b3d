#!/bin/bash
# All bench configurations + rocprofv3 kernel stats and HBM-traffic counter passes for c2 and c3.
# usage: bash tools/jobs/bench_all.sh <tag> [prof]
TAG=${1:-bench}; PROF=$2
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for cfg in c2 c3 argos; do
  python bench.py --config $cfg --steps 10 --warmup 3 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; echo "$cfg rc=$?"
done
python bench.py --config c2 --steps 6 --warmup 2 --captures 8 --no-cpu > $OUT/bench_c2_batch8.json 2>> $OUT/bench_c2.err
if [ -n "$PROF" ]; then
  cd /tmp
  for cfg in c2 c3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o s -- python $R/bench.py --config $cfg --steps 4 --warmup 1 --no-cpu > $OUT/stats_$cfg.log 2>&1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$cfg -o f -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$cfg -o w -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
    cp $(ls $OUT/stats_$cfg/*kernel_stats.csv) $OUT/rocprofv3_kernel_stats_bench_$cfg.csv
    python $R/tools/pmc_traffic.py $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/pmc_hbm_traffic_bench_$cfg.json > $OUT/pmc_$cfg.txt 2>&1
    rm -rf $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/stats_$cfg
  done
  cd $R
fi
python - <<PY
import json
for f in ("bench_c2", "bench_c3", "bench_argos", "bench_c2_batch8"):
    try:
        d = json.loads(open("$OUT/" + f + ".json").read())
        print(f, d["value"], "Msps", d["ms_per_step"], "ms; e2e", d.get("e2e", {}).get("ms"), d.get("e2e", {}).get("runs_ms"), "cli", d.get("e2e_cli", {}).get("seconds"),
              "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("e2e_value"), "parity", d.get("parity_with_cpu_baseline"))
        print("   ", {k: v["ms"] for k, v in d.get("stages", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
for f in $OUT/*.err; do tail -n 3 $f; done
