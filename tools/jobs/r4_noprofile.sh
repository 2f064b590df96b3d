#!/bin/bash
# the c3 step with and without the per-group timing events (profile) -- do the timers cost anything?
python - <<'PY'
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
pdt = importlib.import_module("project-desert-tortoise_amd")
fs = 250000; n = fs * 3600
d_iq = bench.make_capture(pdt, bench.capture_params(pdt, "c3", 1234), n, 32, device=torch.device("cuda", 0), fs=fs)
for prof in (True, False, True, False):
    d = pdt.Demodulator(pdt.MODE_POES, fs, profile=prof).keep_pll(False)
    for _ in range(3): d.demod_device(d_iq.data_ptr(), n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): d.demod_device(d_iq.data_ptr(), n)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 8 * 1e3
    print("profile", prof, "step %.2f ms" % ms, "gpu_ms %.2f" % d.stats().gpu_ms, flush=True)
    d.close()
PY
