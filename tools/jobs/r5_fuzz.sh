#!/bin/bash
# a long fuzz campaign with fresh seeds (GPU path against the oracle, every stage bit-identical); logs under gpurun_out/r5fuzz
# usage: bash tools/jobs/r5_fuzz.sh [seed] [scale]   (scale 1: 600 + 300 + 200 + 80 cases)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5fuzz; mkdir -p $O
export TMPDIR=/tmp
S=${1:-9001}; K=${2:-1}
( time timeout $((1000 * K)) python tests/tools/fuzz.py $((600 * K)) $S ) > $O/fuzz_$S.log 2>&1; tail -3 $O/fuzz_$S.log
( time timeout $((700 * K)) python tests/tools/fuzz_live.py $((300 * K)) $((S+1)) ) > $O/fuzz_live_$((S+1)).log 2>&1; tail -3 $O/fuzz_live_$((S+1)).log
( time timeout $((700 * K)) python tests/tools/fuzz_quality.py $((200 * K)) $((S+2)) ) > $O/fuzz_quality_$((S+2)).log 2>&1; tail -3 $O/fuzz_quality_$((S+2)).log
( time timeout $((900 * K)) python tests/tools/fuzz_segments.py $((80 * K)) $((S+3)) ) > $O/fuzz_segments_$((S+3)).log 2>&1; tail -3 $O/fuzz_segments_$((S+3)).log
