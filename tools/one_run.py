"""One C2-size demodulation with geometry overrides from the environment (for rocprofv3 runs on a GPU box)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
kw = {}
for k in ("pll_block", "pll_warm", "agc_block", "agc_warm"):
    if os.environ.get("PDT_" + k.upper()):
        kw[k] = int(os.environ["PDT_" + k.upper()])
iq = pdt.synth_capture(0, 50000, float(os.environ.get("PDT_SECS", "600")), seed=int(os.environ.get("PDT_SEED", "1234")))
d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=False, **kw)
if os.environ.get("PDT_QUALITY"):          # per-chunk reports (averagePhase EMA after the lock + counts)
    d.keep_quality()
for _ in range(4):
    d.demod(iq)
s = d.stats()
print(kw, "gpu_ms", round(s.gpu_ms, 3), "frames", s.frames, "pll fixes", s.pll_seam_fixes, "agc fixes", s.agc_seam_fixes, "walked", s.gardner_walked)
