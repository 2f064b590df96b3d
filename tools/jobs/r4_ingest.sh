#!/bin/bash
# ingest geometry in the no-overlap form: threads x span x copy streams.  usage: bash tools/jobs/r4_ingest.sh
one() { label=$1; shift
  env "$@" PDT_NO_OVERLAP=1 python bench.py --config c3 --steps 2 --warmup 1 --e2e-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['e2e']; print('$label', 'e2e', e['ms'], [round(x) for x in e['runs_ms']], 'demod_fd', e['split_ms']['demod_fd'], 'gpu', e['gpu_ms'])"; }
for t in 4 6 8 12 16; do one "threads=$t" PDT_INGEST_THREADS=$t; done
for s in 1 2 4; do one "threads=8 streams=$s" PDT_INGEST_THREADS=8 PDT_INGEST_STREAMS=$s; done
for m in 2 4 16 32; do one "threads=8 span=$m" PDT_INGEST_THREADS=8 PDT_INGEST_SPAN_MB=$m; done
