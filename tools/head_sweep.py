"""Head length (in tracking-loop time constants) and PLL block size vs times on C2-size captures (GPU box)."""
import importlib, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    pdt = importlib.import_module("project-desert-tortoise_amd")
    for seed in (1234, 77):
        iq = pdt.synth_capture(0, 50000, 600.0, seed=seed)
        for B in (5000, 2500, 1250):
            d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True, pll_block=B)
            d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
            print(f"taus {os.environ.get('PDT_HEAD_TAUS')} seed {seed} B {B}: fixes {s.pll_seam_fixes} phase {kt['pll_phase'][1]:.2f} acq {kt['pll_acquire'][1]:.2f} "
                  f"head {kt['pll_head'][1]:.2f} fix {kt['pll_fix'][1]:.2f} gpu_ms {s.gpu_ms:.2f} lock {s.lock_sample}", flush=True)
            d.close()
else:
    for t in ("40", "34", "28", "22"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PDT_HEAD_TAUS=t))
