"""k_pll_phase time against the tracking warm-up length and the block length (run on a GPU box): slope = ns per step."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
iq = pdt.synth_capture(0, 50000, 600.0, seed=1234)
for blk, warm in ((2500, 4000), (2500, 8000), (2500, 16000), (2500, 32000), (5000, 16000), (10000, 16000)):
    d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True, pll_block=blk, pll_warm=warm)
    d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
    print(f"block {blk} warm {warm}: pll_phase {kt['pll_phase'][1]:.3f} ms  fixes {s.pll_seam_fixes}  pll_fix {kt['pll_fix'][1]:.3f}", flush=True)
    d.close()
