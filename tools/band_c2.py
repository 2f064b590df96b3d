"""Sweep of the Gardner table band pad: candidates, walked chunks and kernel times (run on a GPU box)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sweep(label, rate, iq, pads):
    for pad in pads:
        d = pdt.Demodulator(pdt.MODE_POES, rate, profile=True, gardner_band_pad=pad)
        d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
        nch = max(1, s.samples // 10000)
        print(f"{label} pad {pad:.5f}: cand {s.gardner_candidates} ({s.gardner_candidates/nch:.0f}/chunk) walked {s.gardner_walked} "
              f"full {s.gardner_full_domain} scout {kt.get('gardner_scout', (0, 0))[1]:.2f} table {kt['gardner_table'][1]:.2f} ms "
              f"chain {kt['gardner_chain'][1]:.2f} gpu_ms {s.gpu_ms:.2f} frames {s.frames}", flush=True)
        d.close()


pads = (1 / 16, 1 / 32, 1 / 64, 1 / 128, 1 / 256)
import wave
w = wave.open(os.path.join(ROOT, "tests/golden/5sec_clip.wav"))
clip = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).reshape(-1, 2)
sweep("clip", 50000, np.tile(clip, (8, 1)), pads)
sweep("synth50k-60s", 50000, pdt.synth_capture(0, 50000, 60.0, seed=7), pads)
sweep("synth250k-12s", 250000, pdt.synth_capture(0, 250000, 12.0, seed=8), pads)
sweep("C2", 50000, pdt.synth_capture(0, 50000, 600.0, seed=1234), pads)
