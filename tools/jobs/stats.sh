#!/bin/bash
# rocprofv3 kernel stats of one bench configuration, readable names. usage: bash tools/jobs/stats.sh <cfg> [extra bench args]
export TMPDIR=/tmp
R=$PWD; CFG=${1:-c2}; shift
OUT=$R/gpurun_out/stats_$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/bench.py --config $CFG --steps 4 --warmup 1 --no-cpu "$@" > $OUT/bench.json 2> $OUT/err.log
cd $R
python - <<PY
import csv, glob, sys, json
sys.path.insert(0, "tools")
from kname import kernel_name
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    print(f"{kernel_name(r['Name'])[:48]:48s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']}%")
d = json.loads(open("$OUT/bench.json").read()); print("walked", d["gardner_walked"], "cand", d["gardner_candidates"], d["ms_per_step"])
PY
