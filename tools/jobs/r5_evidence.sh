#!/bin/bash
# round 5: the ONE evidence set of the build in the tree -- the full GPU suite, the fuzzers, the bench lines of every workload, rocprofv3
# kernel statistics, the two HBM counter passes and one SQ pass of the c3 line, the interleaved end-to-end A/B, the round's probes, the
# C host program's own timing.   usage: bash tools/jobs/r5_evidence.sh <tag> [seed]
TAG=${1:-r5final}; SEED=${2:-501}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 1500 python tests/tools/fuzz.py 300 $SEED > $OUT/fuzz_300_seed$SEED.log 2>&1; tail -1 $OUT/fuzz_300_seed$SEED.log
timeout 900 python tests/tools/fuzz_live.py 200 $((SEED + 100)) > $OUT/fuzz_live_200_seed$((SEED + 100)).log 2>&1; tail -1 $OUT/fuzz_live_200_seed$((SEED + 100)).log
timeout 900 python tests/tools/fuzz_segments.py 60 $((SEED + 200)) > $OUT/fuzz_segments_60_seed$((SEED + 200)).log 2>&1; tail -1 $OUT/fuzz_segments_60_seed$((SEED + 200)).log
( time timeout 1500 python bench.py --gpus 1 ) > $OUT/bench_default_1gpu.json 2> $OUT/bench_default.err; tail -2 $OUT/bench_default.err
for cfg in c2 argos aos weak; do
  timeout 900 python bench.py --config $cfg --steps 10 --warmup 3 --no-secondary > $OUT/bench_${cfg}_1gpu.json 2> $OUT/bench_$cfg.err; echo "$cfg rc=$?"
done
timeout 600 python bench.py --config c2 --steps 6 --warmup 2 --captures 8 --no-cpu --no-secondary > $OUT/bench_c2_batch8_1gpu.json 2>> $OUT/bench_c2.err
timeout 900 python tools/e2e_ab.py 24 plain=PDT_NO_OVERLAP:1 overlapped=PDT_X:0 2>&1 | grep -v amdgpu.ids > $OUT/e2e_ab_interleaved.txt; cat $OUT/e2e_ab_interleaved.txt | cut -c1-60
bash tools/jobs/r5_cli.sh > /dev/null 2>&1; cp gpurun_out/r5/cli_cold.txt $OUT/cli_cold.txt 2>/dev/null
timeout 300 python tools/probes/stage_concurrency.py 2>&1 | grep -v amdgpu.ids > $OUT/stage_concurrency.txt
timeout 300 python tools/probes/dma_ring_beside_kernels.py 2>&1 | grep -v amdgpu.ids > $OUT/dma_ring_beside_kernels.txt
[ -x tools/probes/cold_path_probe ] && timeout 120 ./tools/probes/cold_path_probe > $OUT/cold_path_probe.txt 2>&1
cd /tmp
for cfg in c3 c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o s -- python $R/bench.py --config $cfg --steps 4 --warmup 1 --no-cpu --no-secondary > $OUT/stats_$cfg.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$cfg -o f -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$cfg -o w -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
  cp $(ls $OUT/stats_$cfg/*kernel_stats.csv $OUT/stats_$cfg/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_bench_$cfg.csv
  (cd $R && python tools/pmc_traffic.py $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/pmc_hbm_traffic_bench_$cfg.json > $OUT/pmc_$cfg.txt 2>&1)
  rm -rf $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/stats_$cfg
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -o p -- python $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $OUT/sq_err.log
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
    for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
        wc = v.get("SQ_WAVE_CYCLES", 1) or 1
        print(k[:34], "waves", int(v.get("SQ_WAVES", 0)), {c[3:]: round(x / wc, 3) for c, x in v.items() if c not in ("SQ_WAVE_CYCLES", "SQ_WAVES")})
except Exception as e:
    print("sq pass failed", e)
for f in ("bench_default_1gpu", "bench_c2_1gpu", "bench_argos_1gpu", "bench_aos_1gpu", "bench_weak_1gpu", "bench_c2_batch8_1gpu"):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json") if l.startswith("{")][-1])
        print(f, d["value"], "Msps", d["ms_per_step"], "ms; roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["bound"], "traffic", d["roofline"].get("traffic"),
              "e2e", d.get("e2e", {}).get("ms"), "cli", d.get("e2e_cli", {}).get("seconds"), "cpu", d.get("cpu_baseline", {}).get("value"), "parity", d.get("parity"))
        print("   ", {k: v["ms"] for k, v in d.get("stages", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
rm -rf $OUT/sq
head -12 $OUT/pmc_c3.txt
