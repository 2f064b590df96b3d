"""Per-kernel HBM traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

Collect on a GPU box (separate passes, no trace domains besides --kernel-trace):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu
then:  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r3/pmc_hbm_traffic_bench_c3.json
(run it on the GPU box, in the same call: the file records the build tag of the libpdt.so that was measured, and bench.py
quotes `traffic` only from a file whose tag equals the running library's)

FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced streaming reads -- it is doubled here ("FETCH_corrected_KB"); WRITE_SIZE is taken
as reported.  Values are averages per launch.
"""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import kernel_name
from collections import defaultdict


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = kernel_name(r["Kernel_Name"])
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


fetch, n1 = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
    if k.startswith("__amd"):
        continue
    rows.append({"kernel": k, "launches": n1[k], "FETCH_SIZE_KB": round(fetch[k], 3), "FETCH_corrected_KB": round(2 * fetch[k], 3),
                 "WRITE_SIZE_KB": round(write.get(k, 0.0), 3), "hbm_bytes": round((2 * fetch[k] + write.get(k, 0.0)) * 1024)})
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
tag = importlib.import_module("project-desert-tortoise_amd").build_tag()
json.dump({"build": tag, "kernels": rows}, open(sys.argv[3], "w"), indent=1)
for r in rows:
    print(f"{r['kernel'][:48]:48s} fetch*2 {r['FETCH_corrected_KB'] / 1024:9.1f} MiB  write {r['WRITE_SIZE_KB'] / 1024:9.1f} MiB")
