"""PLL block / warm-up geometry vs kernel times on the C2 workload (run on a GPU box)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
iq = pdt.synth_capture(0, 50000, float(os.environ.get("PDT_SECS", "600")), seed=1234)
for B, W in ((5000, 15000), (2500, 15000), (1250, 15000), (5000, 12500), (2500, 12500), (2500, 10000), (1250, 10000), (1250, 7500)):
    d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True, pll_block=B, pll_warm=W)
    d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
    print(f"B {B} W {W}: blocks {s.pll_blocks} fixes {s.pll_seam_fixes} phase {kt['pll_phase'][1]:.2f} acquire {kt['pll_acquire'][1]:.2f} "
          f"fix {kt['pll_fix'][1]:.2f} gpu_ms {s.gpu_ms:.2f} frames {s.frames}", flush=True)
    d.close()
