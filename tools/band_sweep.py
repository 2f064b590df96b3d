"""Gardner table band pad vs walked chunks / table time."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
rng = np.random.default_rng(3)
clip = pdt.read_wav(os.path.join(ROOT, "tests/golden/5sec_clip.wav"))[1]
noisy = (pdt.synth_capture(0, 50000, 60.0, seed=9).astype(np.int32) + rng.integers(-2500, 2500, size=(3000000, 2))).clip(-32768, 32767).astype(np.int16)
caps = (("synth50k_120s", 50000, pdt.synth_capture(0, 50000, 120.0, seed=1234)), ("synth50k_noisy_60s", 50000, noisy),
        ("clip x6", 50000, np.concatenate([clip] * 6)), ("synth250k_24s", 250000, pdt.synth_capture(0, 250000, 24.0, seed=7)))
for name, fs, iq in caps:
    ref = None
    for pad in (0.5, 0.375, 0.25, 0.125, 0.0625):
        d = pdt.Demodulator(pdt.MODE_POES, fs, profile=True, gardner_band_pad=pad)
        d.demod(iq); s = d.stats(); kt = d.kernel_times(); t = d.text()
        ref = ref or t
        print(f"{name:20s} pad {pad:6.4f}: cand/chunk {s.gardner_candidates / max(1, s.samples * s.interp // (10000 * s.interp)):7.0f} walked {s.gardner_walked:4d} full {s.gardner_full_domain:3d} "
              f"table {kt['gardner_table'][1]:6.2f} chain {kt['gardner_chain'][1]:6.2f} ms  same_output {t == ref}")
        d.close()
