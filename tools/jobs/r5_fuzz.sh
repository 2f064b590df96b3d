#!/bin/bash
# a long fuzz campaign with fresh seeds (GPU path against the oracle, every stage bit-identical); logs under gpurun_out/r5fuzz
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5fuzz; mkdir -p $O
export TMPDIR=/tmp
S=${1:-9001}
( time timeout 1000 python tests/tools/fuzz.py 600 $S ) > $O/fuzz_$S.log 2>&1; tail -3 $O/fuzz_$S.log
( time timeout 700 python tests/tools/fuzz_live.py 300 $((S+1)) ) > $O/fuzz_live_$((S+1)).log 2>&1; tail -3 $O/fuzz_live_$((S+1)).log
( time timeout 700 python tests/tools/fuzz_quality.py 200 $((S+2)) ) > $O/fuzz_quality_$((S+2)).log 2>&1; tail -3 $O/fuzz_quality_$((S+2)).log
( time timeout 900 python tests/tools/fuzz_segments.py 80 $((S+3)) ) > $O/fuzz_segments_$((S+3)).log 2>&1; tail -3 $O/fuzz_segments_$((S+3)).log
