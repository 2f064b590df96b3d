#!/bin/bash
# GPU-box job: GPU test suite, the bench line, the batched mode.  usage: bash tools/jobs/check.sh <tag> [pytest args]
TAG=${1:-run}; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -x -q "$@" > $OUT/gputests.log 2>&1; tail -15 $OUT/gputests.log
python bench.py --steps 10 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python bench.py --steps 6 --warmup 2 --captures 8 --no-cpu > $OUT/bench_c2_batch8.json 2>> $OUT/bench.err
python - <<PY
import json
for f in ("bench_c2", "bench_c2_batch8"):
    try:
        d = json.loads(open("$OUT/" + f + ".json").read())
        print(f, d["value"], "Msps", d["ms_per_step"], "ms", {k: v["ms"] for k, v in d.get("stages", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
