#!/bin/bash
# c3 end to end: host threads and span size of the ingest (PDT_INGEST_THREADS, PDT_INGEST_SPAN_MB)
for cfg in "8 8" "8 4" "12 8" "6 8" "8 12" "12 4"; do
  set -- $cfg
  export PDT_INGEST_THREADS=$1 PDT_INGEST_SPAN_MB=$2
  echo "threads $1 span $2 MiB"; python bench.py --config ${CFG:-c3} --steps 3 --warmup 1 --e2e-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print(d['ms_per_step'], e['ms'], e['runs_ms'], e['split_ms'])"
done
