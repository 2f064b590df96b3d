#!/bin/bash
# ARGOS path: its GPU tests, then the configs[3] bench line (stages).  usage: bash tools/jobs/argos.sh <tag>
TAG=${1:-argos}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "argos or mixed or stream or long" > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 300 python bench.py --config argos --steps 10 --warmup 3 > $OUT/bench_argos.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_argos.json").read())
print(d["value"], "Msps", d["ms_per_step"], "ms", "fixes", d.get("pll_seam_fixes"), d.get("agc_seam_fixes"), "parity", d.get("parity_with_cpu_baseline"), {k: v["ms"] for k, v in d.get("stages", {}).items()})
PY
