#!/bin/bash
# per-call durations of the k_pll_fix launches (c3)
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/r3trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $OUT/err.log
cd $R
python - <<PY
import csv, glob, sys
sys.path.insert(0, "tools")
from kname import kernel_name
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fx = [(kernel_name(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X") or r.get("Grid_Size")) for r in rows]
last = [i for i, x in enumerate(fx) if x[0].startswith("k_static_gain")][-1]
t0 = int(rows[last]["Start_Timestamp"])
for i in range(last, len(fx)):
    n, us, g = fx[i]
    if "pll" in n or "mix" in n or "agc" in n:
        print(f"{(int(rows[i]['Start_Timestamp']) - t0) / 1e3:9.1f} us  {n[:40]:40s} {us:9.1f} us grid {g}")
PY
rm -rf $OUT/t
