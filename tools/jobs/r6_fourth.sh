#!/bin/bash
# round 6, fourth GPU job: the acquisition in batches of 128 (two samples per lane), run_capture as Chain<T> -- suite, fuzz, the pre-lock workloads
TAG=${1:-r6d}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -3; grep -n "^FAILED\|^E  " $OUT/pytest_gpu.log | head -20
timeout 1200 python tests/tools/fuzz.py 150 601 > $OUT/fuzz_150_seed601.log 2>&1; tail -1 $OUT/fuzz_150_seed601.log
timeout 600 python tests/tools/fuzz_segments.py 30 602 > $OUT/fuzz_segments_30_seed602.log 2>&1; tail -1 $OUT/fuzz_segments_30_seed602.log
for cfg in pass aos; do
  timeout 900 python bench.py --config $cfg --steps 3 --warmup 1 --no-secondary > $OUT/bench_${cfg}_1gpu.json 2> $OUT/bench_$cfg.err; echo "$cfg rc=$?"; tail -2 $OUT/bench_$cfg.err | cut -c1-400
done
timeout 900 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu > $OUT/bench_c3_resident_1gpu.json 2> $OUT/bench_c3.err
python - <<PY
import json
for f in ("bench_pass_1gpu", "bench_aos_1gpu", "bench_c3_resident_1gpu"):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json") if l.startswith("{")][-1])
        par = d.get("parity", {})
        bad = [k for k, v in par.items() if v is False or (isinstance(v, list) and v and isinstance(v[0], bool) and not all(v))]
        print(f, d["value"], "Msps", d["ms_per_step"], "ms; e2e", d.get("value_e2e"), d.get("ms_e2e"), "parity bad:", bad, "lock", d.get("lock_sample"))
        print("   ", {k: v["ms"] for k, v in d.get("stages", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
