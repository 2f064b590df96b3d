#!/bin/bash
# the c3 step's timeline: where the GPU idles between kernels (rocprofv3 kernel trace of three steps)
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/gaps; rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o k -- python $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
cp $(ls $OUT/t/*kernel_trace.csv | head -1) $OUT/kernel_trace.csv; rm -rf $OUT/t
