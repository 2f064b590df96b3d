"""Resident step with and without the per-chunk reports (pdt_keep_quality), c2 and c3 sizes; PDT_QUALITY_INLINE=1 for the old order."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pdt = importlib.import_module("project-desert-tortoise_amd")
import bench
for cfg, fs, n in (("c2", 50000, 30000000), ("c3", 250000, 900000000)):
    p = bench.capture_params(pdt, cfg, 1234)
    d_iq = bench.make_capture(pdt, p, n, 16, device="cuda:0")
    for q in (False, True):
        with pdt.Demodulator(pdt.MODE_POES, fs, profile=True).keep_pll(False) as d:
            if q:
                d.keep_quality()
            ms = []
            for i in range(4):
                d.demod_device(d_iq.data_ptr(), n)
                ms.append(round(d.stats().gpu_ms, 3))
            kt = d.kernel_times()
            print(cfg, "quality", q, "inline", bool(os.environ.get("PDT_QUALITY_INLINE")), "gpu_ms", ms, "quality group", round(kt.get("quality", (1, 0))[1] / max(kt.get("quality", (1, 0))[0], 1), 3))
    del d_iq
    torch.cuda.empty_cache()
