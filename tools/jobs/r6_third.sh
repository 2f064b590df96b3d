#!/bin/bash
# round 6, third GPU job: two-pass (O_DIRECT) ingest + NUMA placement, packed arithmetic in the interpolating FIR, the rows' exit histogram
TAG=${1:-r6c}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -3; grep -n "^FAILED\|^E  " $OUT/pytest_gpu.log | head -20
for cfg in i8 c2h c2; do
  timeout 900 python bench.py --config $cfg --steps 5 --warmup 2 --no-secondary --no-cpu > $OUT/bench_${cfg}_1gpu.json 2> $OUT/bench_$cfg.err; echo "$cfg rc=$?"
done
timeout 600 python tools/span_hist.py c3 > $OUT/gardner_row_exits_c3.txt 2>&1; cat $OUT/gardner_row_exits_c3.txt | grep -v amdgpu
timeout 600 python tools/span_hist.py c2h > $OUT/gardner_row_exits_c2h.txt 2>&1; cat $OUT/gardner_row_exits_c2h.txt | grep -v amdgpu | head -4
# where do the box's file systems stand: tmpfs (/dev/shm) and the disk under /tmp; NUMA layout
( df -h /dev/shm /tmp | cat; cat /sys/devices/system/node/online; ls /sys/devices/system/node/ | head; nproc; free -g | head -2 ) > $OUT/host.txt 2>&1; cat $OUT/host.txt
# the c3 file -> frame file path: tmpfs buffered (today's), with / without NUMA binding; disk file cold (O_DIRECT) and cached
timeout 900 python tools/e2e_ab.py 8 auto=PDT_X:0 numa_on=PDT_INGEST_NUMA:1 numa_off=PDT_INGEST_NUMA:0 2>&1 | grep -v amdgpu.ids > $OUT/e2e_ab_numa.txt; cat $OUT/e2e_ab_numa.txt | cut -c1-200
AB_DIR=/tmp AB_SECONDS=900 timeout 900 python tools/e2e_ab.py 4 cached=PDT_X:0 cold=AB_EVICT:1 cold_buffered=AB_EVICT:1,PDT_INGEST_DIRECT:0 2>&1 | grep -v amdgpu.ids > $OUT/e2e_ab_disk.txt; cat $OUT/e2e_ab_disk.txt | cut -c1-200
( time timeout 1500 python bench.py --gpus 1 ) > $OUT/bench_default_1gpu.json 2> $OUT/bench_default.err; tail -2 $OUT/bench_default.err | cut -c1-300
python - <<PY
import json
for f in ("bench_i8_1gpu", "bench_c2h_1gpu", "bench_c2_1gpu", "bench_default_1gpu"):
    try:
        d = json.loads([l for l in open("$OUT/" + f + ".json") if l.startswith("{")][-1])
        print(f, d["value"], "Msps", d["ms_per_step"], "ms; e2e", d.get("value_e2e"), d.get("ms_e2e"), "in-process", d.get("e2e", {}).get("ms"), "cli", d.get("e2e_cli", {}).get("seconds"))
        print("   ", {k: v["ms"] for k, v in d.get("stages", {}).items()})
        if "e2e_multi" in d: print("    e2e_multi", {k: d["e2e_multi"].get(k) for k in ("ms", "passes_ms", "until_last_gpu_ms", "gather_ms", "write_ms", "per_gpu")})
        if "secondary" in d: print("    secondary", {k: (v.get("value"), v.get("ms_per_step"), v.get("error")) for k, v in d["secondary"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
