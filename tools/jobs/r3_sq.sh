#!/bin/bash
# SQ counters of the big kernels on the c3 line (one pass, 8 SQ slots).  usage: bash tools/jobs/r3_sq.sh <tag>
TAG=${1:-r3sq}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p -o p -- python $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $OUT/err.log
cd $R
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "tools")
from kname import kernel_name
f = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = kernel_name(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {}
for k in acc:
    out[k] = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
json.dump(out, open("$OUT/sq_counters_bench_c3.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k[:44]:44s} waves {v.get('SQ_WAVES',0):9.0f} wavecyc {wc:14.0f} wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active {v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} valu {v.get('SQ_INSTS_VALU',0):13.0f} salu {v.get('SQ_INSTS_SALU',0):13.0f} lds {v.get('SQ_INSTS_LDS',0):12.0f}")
PY
rm -rf $OUT/p
