#!/bin/bash
for cfg in "1 16 2" "2 16 2" "4 16 2" "4 16 1" "4 8 4"; do
  set -- $cfg
  export PDT_INGEST_STREAMS=$1 PDT_INGEST_THREADS=$2 PDT_INGEST_SPAN_MB=$3
  echo "streams $1 threads $2 span $3 MiB"; python bench.py --config c2 --steps 3 --warmup 1 --e2e-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print(d['ms_per_step'], e['ms'], e['runs_ms'], e['split_ms'])"
done
