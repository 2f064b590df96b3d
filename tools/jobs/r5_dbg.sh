#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
python - <<'PY' 2>&1 | tee gpurun_out/r5/dbg_stream.log
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
pdt = importlib.import_module("project-desert-tortoise_amd")
rate, iq = pdt.read_wav("tests/golden/5sec_clip.wav")
for env in ({}, {"PDT_SEG_PLAIN": "1"}):
    os.environ.update(env)
    with pdt.Demodulator(pdt.MODE_POES, rate) as d:
        d.stream_begin()
        tot = 0
        for i in range(0, len(iq), 2400):
            new = d.stream_push(iq[i:i + 2400])
            tot += len(new)
            s = d.stats()
            if i % 24000 == 0 or len(new):
                print(env, i, "new", len(new), "sym", s.symbols, "bits", s.bits, "frames", s.frames, "lock", s.lock_sample, "norm", s.norm_factor)
        tail = d.stream_end()
        print(env, "total", tot + len(tail), d.stats().frames)
    for k in env: os.environ.pop(k)
PY
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary > /tmp/prof_bench.log 2>&1
cd $R
f=$(ls /tmp/prof/*kernel_stats.csv /tmp/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" gpurun_out/r5/dbg_kernel_stats.csv
python - <<PY
import csv, sys
sys.path.insert(0, "tools")
from kname import kernel_name
for r in list(csv.DictReader(open("gpurun_out/r5/dbg_kernel_stats.csv")))[:25]:
    print(kernel_name(r["Name"])[:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
