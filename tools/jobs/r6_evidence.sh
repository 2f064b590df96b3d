#!/bin/bash
# round 6: the ONE evidence set of the build in the tree -- the full GPU suite, the fuzzers, the bench line of every workload (the driver's
# default line with its secondary workloads; pass / aos / weak / i8 / c2h; the batched modes), rocprofv3 kernel statistics (c3, pass, i8, c2h),
# the two HBM counter passes (c3, i8, c2h) and one SQ pass (c3), the rows' exit histogram, the C host program's own timing.
# (The fuzzers run under faulthandler and are ended with SIGABRT on time-out: a stall -- one was seen once in round 6, in fuzz.py 150 601
# behind case 23, and never again in five repetitions -- leaves the Python stack of the call it sits in.)
#   usage: bash tools/jobs/r6_evidence.sh <tag> [seed]
TAG=${1:-r6final}; SEED=${2:-611}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT/logs
( time timeout 2400 python -m pytest tests -m gpu -q ) > $OUT/logs/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/logs/pytest_gpu.log | tail -2
timeout -s ABRT 1500 python -X faulthandler tests/tools/fuzz.py 300 $SEED > $OUT/logs/fuzz_300_seed$SEED.log 2>&1; tail -1 $OUT/logs/fuzz_300_seed$SEED.log
timeout -s ABRT 900 python -X faulthandler tests/tools/fuzz_live.py 200 $((SEED + 100)) > $OUT/logs/fuzz_live_200_seed$((SEED + 100)).log 2>&1; tail -1 $OUT/logs/fuzz_live_200_seed$((SEED + 100)).log
timeout -s ABRT 900 python -X faulthandler tests/tools/fuzz_segments.py 60 $((SEED + 200)) > $OUT/logs/fuzz_segments_60_seed$((SEED + 200)).log 2>&1; tail -1 $OUT/logs/fuzz_segments_60_seed$((SEED + 200)).log
timeout -s ABRT 900 python -X faulthandler tests/tools/fuzz_quality.py 100 $((SEED + 300)) > $OUT/logs/fuzz_quality_100_seed$((SEED + 300)).log 2>&1; tail -1 $OUT/logs/fuzz_quality_100_seed$((SEED + 300)).log
( time timeout 1800 python bench.py --gpus 1 ) > $OUT/bench_default_1gpu.json 2> $OUT/logs/bench_default.err; tail -2 $OUT/logs/bench_default.err | cut -c1-300
for cfg in c2 argos aos weak pass i8 c2h; do
  timeout 900 python bench.py --config $cfg --steps 5 --warmup 2 --no-secondary > $OUT/bench_${cfg}_1gpu.json 2> $OUT/logs/bench_$cfg.err; echo "$cfg rc=$?"
done
timeout 600 python bench.py --config c2 --steps 6 --warmup 2 --captures 8 --no-secondary > $OUT/bench_c2_batch8_1gpu.json 2>> $OUT/logs/bench_c2.err
for nc in 32 64; do
  timeout 900 python bench.py --config argos --captures $nc --steps 5 --warmup 2 --no-secondary > $OUT/bench_argos_batch${nc}_1gpu.json 2> $OUT/logs/bench_argos_batch$nc.err; echo "argos x$nc rc=$?"
done
timeout 900 python bench.py --config pass --captures 16 --steps 2 --warmup 1 --no-secondary > $OUT/bench_pass_batch16_1gpu.json 2> $OUT/logs/bench_pass_batch16.err; echo "pass x16 rc=$?"
timeout 600 python tools/span_hist.py c3 2>&1 | grep -v amdgpu.ids > $OUT/gardner_row_exits_c3.txt
timeout 600 python tools/span_hist.py c2h 2>&1 | grep -v amdgpu.ids > $OUT/gardner_row_exits_c2h.txt
bash tools/jobs/cli_cold.sh > /dev/null 2>&1; cp gpurun_out/cli/cli_cold.txt $OUT/cli_cold.txt 2>/dev/null
cd /tmp
for cfg in c3 pass i8 c2h; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o s -- python $R/bench.py --config $cfg --steps 4 --warmup 1 --no-cpu --no-secondary > $OUT/logs/stats_$cfg.log 2>&1
  cp $(ls $OUT/stats_$cfg/*kernel_stats.csv $OUT/stats_$cfg/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_bench_$cfg.csv
  rm -rf $OUT/stats_$cfg
done
for cfg in c3 i8 c2h; do
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$cfg -o f -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$cfg -o w -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
  (cd $R && python tools/pmc_traffic.py $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/pmc_hbm_traffic_bench_$cfg.json > $OUT/pmc_$cfg.txt 2>&1)
  rm -rf $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -o p -- python $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $OUT/logs/sq_err.log
cd $R
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "tools")
from kname import kernel_name
try:
    f = glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = kernel_name(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    out = {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}
    sys.path.insert(0, ".")
    import importlib
    json.dump({"build": importlib.import_module("project-desert-tortoise_amd").build_tag(), "kernels": out}, open("$OUT/sq_counters_bench_c3.json", "w"), indent=1)
except Exception as e:
    print("sq pass failed", e)
for f in ("bench_default_1gpu", "bench_c2_1gpu", "bench_argos_1gpu", "bench_aos_1gpu", "bench_weak_1gpu", "bench_pass_1gpu", "bench_i8_1gpu", "bench_c2h_1gpu", "bench_c2_batch8_1gpu", "bench_argos_batch32_1gpu", "bench_argos_batch64_1gpu", "bench_pass_batch16_1gpu"):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json") if l.startswith("{")][-1])
        par = d.get("parity", {})
        bad = [k for k, v in par.items() if v is False or (isinstance(v, list) and v and isinstance(v[0], bool) and not all(v))]
        print(f, d["value"], "Msps", d["ms_per_step"], "ms; roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["bound"], "traffic", d["roofline"].get("traffic"),
              "e2e", d.get("value_e2e"), d.get("ms_e2e"), "in-process", d.get("e2e", {}).get("ms"), "cli", d.get("e2e_cli", {}).get("seconds"), "cpu", d.get("cpu_baseline", {}).get("value"), "parity bad", bad)
        print("   ", {k: v["ms"] for k, v in d.get("stages", {}).items()})
        if "secondary" in d: print("    secondary", {k: (v.get("value"), v.get("ms_per_step"), v.get("error")) for k, v in d["secondary"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
rm -rf $OUT/sq
head -12 $OUT/pmc_c3.txt; head -8 $OUT/pmc_i8.txt; head -8 $OUT/pmc_c2h.txt
