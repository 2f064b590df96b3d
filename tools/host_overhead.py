"""Wall time per demodulation vs the device time between the first and last kernel (run on a GPU box)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
iq = pdt.synth_capture(0, 50000, 600.0, seed=1234)
dev = torch.from_numpy(iq).to("cuda:0")
for prof in (False, True):
    d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=prof)
    for _ in range(3):
        d.demod_device(dev.data_ptr(), len(iq))
    t0 = time.perf_counter(); K = 20; g = 0.0
    for _ in range(K):
        d.demod_device(dev.data_ptr(), len(iq)); g += d.stats().gpu_ms
    dt = (time.perf_counter() - t0) / K * 1e3
    print(f"profile={prof}: wall {dt:.3f} ms per call, device {g / K:.3f} ms, host-only {dt - g / K:.3f} ms")
    d.close()
