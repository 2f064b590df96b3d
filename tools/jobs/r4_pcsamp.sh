#!/bin/bash
# Round 4: where do the walkers' idle clocks go?  rocprofv3 PC sampling of the c3 line (stochastic if the box supports it,
# host-trap otherwise), reduced on the box to a per-kernel, per-instruction histogram with the stall reasons.
# usage: bash tools/jobs/r4_pcsamp.sh [tag] [bench args...]
TAG=${1:-r4pc}; shift
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS=${@:---config c3 --steps 2 --warmup 1 --no-cpu --no-secondary}
cd /tmp
for METHOD in stochastic host_trap; do
  if [ $METHOD = stochastic ]; then UNIT=cycles; INT=${PC_INT:-65536}; else UNIT=time; INT=${PC_INT_US:-50}; fi
  rm -rf /tmp/pcs_$METHOD
  timeout 900 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INT \
      --output-format csv -d /tmp/pcs_$METHOD -o pc -- python $R/bench.py $ARGS > $OUT/bench_$METHOD.log 2> $OUT/err_$METHOD.log
  echo "$METHOD rc=$?" >> $OUT/status.txt
  find /tmp/pcs_$METHOD -type f | xargs ls -la >> $OUT/status.txt 2>&1
  F=$(find /tmp/pcs_$METHOD -name "*pc_sampling*$METHOD*.csv" | head -1)
  [ -z "$F" ] && F=$(find /tmp/pcs_$METHOD -name "*pc_sampling*.csv" | head -1)
  if [ -n "$F" ] && [ $(wc -l < "$F") -gt 10 ]; then
    head -5 "$F" > $OUT/head_$METHOD.txt
    KT=$(find /tmp/pcs_$METHOD -name "*kernel_trace.csv" | head -1)
    python $R/tools/pc_hist.py "$F" "$KT" $OUT/pc_hist_$METHOD.json > $OUT/pc_hist_$METHOD.txt 2>> $OUT/err_$METHOD.log
    break
  fi
done
tail -3 $OUT/err_*.log
cat $OUT/status.txt | tail -20
