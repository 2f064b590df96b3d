#!/bin/bash
# kernel traces of the single-capture bench and the batched mode. usage: bash tools/jobs/trace.sh <tag>
TAG=${1:-trace}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t1 -o c2 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu > $OUT/t1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t8 -o b8 -- python $R/bench.py --steps 2 --warmup 1 --captures 8 --no-cpu > $OUT/t8.log 2>&1
cd $R
python tools/trace_queues.py $(ls $OUT/t8/*kernel_trace.csv) 60 > $OUT/t8_queues.txt 2>&1
head -30 $(ls $OUT/t1/*kernel_stats.csv)
