#!/bin/bash
# second SQ pass on the c3 line: what the issue stalls of the walkers are
TAG=${1:-r3sq2}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p -o p -- python $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $OUT/err.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/q -o q -- python $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>> $OUT/err.log
cd $R
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "tools")
from kname import kernel_name
out = collections.defaultdict(dict)
for d in ("p", "q"):
    f = glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = kernel_name(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        for c in acc[k]:
            out[k][c + ("" if d == "p" or c != "SQ_WAVE_CYCLES" else "_q")] = acc[k][c] / cnt[k][c]
json.dump(out, open("$OUT/sq2_counters_bench_c3.json", "w"), indent=1)
for k in ("k_agc_block<float>", "k_pll_phase<float, false>", "k_gardner_span_walk<2048>", "k_gardner_emit_rest<512>", "k_mix_fir<26, 0>", "k_pll_head<float, false, true>"):
    v = out.get(k, {})
    wc = v.get("SQ_WAVE_CYCLES", 1) or 1
    print(k[:30], {c: round(x / wc, 3) if c.startswith(("SQ_ACTIVE", "SQ_WAIT", "SQ_INST_CYCLES", "SQ_VMEM", "SQ_INST_LEVEL", "SQ_LDS")) else int(x) for c, x in v.items()})
PY
rm -rf $OUT/p $OUT/q
