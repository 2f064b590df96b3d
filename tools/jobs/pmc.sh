#!/bin/bash
# SQ counters of one kernel (regex $1) on the c2 bench. usage: bash tools/jobs/pmc.sh k_pll_phase "SQ_WAVE_CYCLES SQ_WAIT_ANY ..."
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT -o p -- python $R/bench.py --config ${CFG:-c2} --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    if "$1" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k, v in acc.items(): print(k, v[0] / cnt[k])
PY
