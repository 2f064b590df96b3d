#!/bin/bash
# c3 end to end: overlapped ingest (segments) vs not
for cfg in "no" "4" "6" "8" "12"; do
  unset PDT_NO_OVERLAP PDT_OVERLAP_SEGMENTS
  if [ $cfg = no ]; then export PDT_NO_OVERLAP=1; else export PDT_OVERLAP_SEGMENTS=$cfg; fi
  echo "overlap segments $cfg"; python bench.py --config c3 --steps 3 --warmup 1 --e2e-only 2>gpurun_out/c3_overlap_$cfg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print(d['ms_per_step'], e['ms'], e['runs_ms'], e['split_ms'], e['text_identical_to_resident_run'])"
  tail -2 gpurun_out/c3_overlap_$cfg.err | grep -v amdgpu
done
