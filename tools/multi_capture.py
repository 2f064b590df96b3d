"""Throughput with M captures demodulated concurrently on one GPU (M contexts, M host threads; run on a GPU box)."""
import importlib, os, sys, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
n_sec = float(os.environ.get("PDT_SECS", "600"))
iq = pdt.synth_capture(0, 50000, n_sec, seed=1234)
n = len(iq)
dev = torch.from_numpy(iq.view(np.int16).reshape(-1).copy()).to("cuda:0")
torch.cuda.synchronize()
for M in (1, 2, 4, 8, 16):
    ds = [pdt.Demodulator(pdt.MODE_POES, 50000) for _ in range(M)]
    for d in ds:
        d.demod_device(dev.data_ptr(), n)          # warm-up: allocations
    steps = 6
    def work(d):
        for _ in range(steps):
            d.demod_device(dev.data_ptr(), n)
    t0 = time.time()
    th = [threading.Thread(target=work, args=(d,)) for d in ds]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.time() - t0
    t0 = time.time()
    for _ in range(steps):
        pdt.demod_batch(ds, [dev.data_ptr()] * M, [n] * M)
    dtb = time.time() - t0
    print(f"M={M}: batch entry point {M * steps * n / dtb / 1e6:.0f} Msamples/s ({dtb / steps * 1e3:.2f} ms per batch)", flush=True)
    fr = [d.stats().frames for d in ds]
    print(f"M={M}: {M * steps * n / dt / 1e6:.0f} Msamples/s  ({dt / steps * 1e3:.2f} ms per round of {M} captures) frames {fr[0]} gpu_ms {ds[0].stats().gpu_ms:.2f}", flush=True)
    for d in ds:
        d.close()
