"""AGC warm-up length (in gain time constants) vs seam repairs and kernel time (run on a GPU box)."""
import importlib, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, wave
    pdt = importlib.import_module("project-desert-tortoise_amd")
    w = wave.open(os.path.join(ROOT, "tests/golden/5sec_clip.wav"))
    clip = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).reshape(-1, 2)
    cases = [("clip x8", 0, 50000, np.tile(clip, (8, 1)), {}), ("synth 60 s", 0, 50000, pdt.synth_capture(0, 50000, 60.0, seed=7), {}),
             ("C2", 0, 50000, pdt.synth_capture(0, 50000, 600.0, seed=1234), {}),
             ("C2 B/2", 0, 50000, pdt.synth_capture(0, 50000, 600.0, seed=1234), {"agc_block": 4688}),
             ("C2 B/4", 0, 50000, pdt.synth_capture(0, 50000, 600.0, seed=1234), {"agc_block": 2344}),
             ("argos", 1, 32000, pdt.synth_capture(1, 32000, 30.0, seed=5), {})]
    for label, mode, rate, iq, kw in cases:
        d = pdt.Demodulator(mode, rate, profile=True, **kw)
        d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
        print(f"K={os.environ.get('PDT_AGC_K')} {label}: agc blocks {s.agc_blocks} fixes {s.agc_seam_fixes} agc_block {kt['agc_block'][1]:.3f} ms "
              f"agc_fix {kt['agc_fix'][1]:.3f} ms gpu_ms {s.gpu_ms:.2f} frames {s.frames}", flush=True)
        d.close()
else:
    for K in ("14", "10", "8", "6", "5", "4", "3"):
        env = dict(os.environ, PDT_AGC_K=K)
        subprocess.run([sys.executable, __file__, "child"], env=env)
