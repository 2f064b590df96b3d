"""AGC block size vs kernel times on C2-size captures (run on a GPU box)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
for seed in (1234, 77):
    iq = pdt.synth_capture(0, 50000, 600.0, seed=seed)
    for B in (9376, 6252, 4688, 3128, 2344, 1564):
        d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True, agc_block=B)
        d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
        print(f"seed {seed} agc B {B}: blocks {s.agc_blocks} fixes {s.agc_seam_fixes} agc_block {kt['agc_block'][1]:.3f} agc_fix {kt['agc_fix'][1]:.3f} gpu_ms {s.gpu_ms:.2f}", flush=True)
        d.close()
