#!/bin/bash
# the c3 step with the per-chunk reports on (what bin/demodPOES runs) against the benched step.  usage: bash tools/jobs/r4_quality.sh
export TMPDIR=/tmp
python - <<'PY'
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
pdt = importlib.import_module("project-desert-tortoise_amd")
fs = 250000; n = fs * 3600
d_iq = bench.make_capture(pdt, bench.capture_params(pdt, "c3", 1234), n, 32, device=torch.device("cuda", 0), fs=fs)
for q in (0, 1):
    d = pdt.Demodulator(pdt.MODE_POES, fs, profile=True).keep_pll(False)
    if q: d.keep_quality(True)
    for _ in range(3): d.demod_device(d_iq.data_ptr(), n)
    t0 = time.perf_counter()
    for _ in range(5): d.demod_device(d_iq.data_ptr(), n)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    kt = d.kernel_times()
    print("quality", q, "step %.2f ms" % ms, {k: round(v[1], 3) for k, v in kt.items() if k in ("quality", "mix_fir", "agc_block", "gardner_table", "gardner")}, flush=True)
    d.close()
PY
