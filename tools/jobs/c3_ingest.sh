#!/bin/bash
# c3 / c2 end to end: copy streams, host threads and span size of the ingest (PDT_INGEST_STREAMS / _THREADS / _SPAN_MB)
for cfg in "1 16 8" "2 16 8" "4 16 8" "2 8 8" "4 8 16" "2 16 4"; do
  set -- $cfg
  export PDT_INGEST_STREAMS=$1 PDT_INGEST_THREADS=$2 PDT_INGEST_SPAN_MB=$3
  echo "streams $1 threads $2 span $3 MiB"; python bench.py --config ${CFG:-c3} --steps 3 --warmup 1 --e2e-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print(d['ms_per_step'], e['ms'], e['runs_ms'], e['split_ms'])"
done
