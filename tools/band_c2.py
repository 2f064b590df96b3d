"""Sweep of the Gardner candidate pad: candidates, walked chunks and kernel times (run on a GPU box)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")


def sweep(label, rate, iq, pads):
    for pad in pads:
        d = pdt.Demodulator(pdt.MODE_POES, rate, profile=True, gardner_band_pad=pad)
        d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
        nch = max(1, s.samples // 10000)
        print(f"{label} pad {pad:.5f}: cand {s.gardner_candidates} ({s.gardner_candidates/nch:.0f}/chunk) walked {s.gardner_walked} "
              f"full {s.gardner_full_domain} table {kt['gardner_table'][1]:.2f} ms "
              f"chain {kt['gardner_chain'][1]:.2f} gpu_ms {s.gpu_ms:.2f} frames {s.frames}", flush=True)
        d.close()


pads = [float(x) for x in sys.argv[1:]] or [1 / 4, 3 / 16, 1 / 8, 3 / 32, 1 / 16]
for seed in (1234, 77, 4242):
    sweep(f"C2 seed {seed}", 50000, pdt.synth_capture(0, 50000, 600.0, seed=seed), pads)
