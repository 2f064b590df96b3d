#!/bin/bash
# Long parity evidence on the GPU box: fuzz sweep (300 cases, seed $2), configs[2] at full size against the reference's CPU
# objects, the 5-minute ARGOS capture against them, the noise sweep.  usage: bash tools/jobs/evidence.sh <tag> [seed]
TAG=${1:-evidence}; SEED=${2:-77}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python tests/tools/fuzz.py 300 $SEED > $OUT/fuzz_300_seed$SEED.log 2>&1; tail -2 $OUT/fuzz_300_seed$SEED.log
PDT_SECS=3600 PDT_RATE=250000 timeout 900 python tests/tools/c3_check.py > $OUT/c3_full_900M.log 2>&1; tail -3 $OUT/c3_full_900M.log
timeout 600 python tests/tools/argos_long.py > $OUT/argos_long.log 2>&1; tail -4 $OUT/argos_long.log
timeout 900 python tests/tools/noise_sweep.py > $OUT/noise_sweep.log 2>&1; tail -10 $OUT/noise_sweep.log
