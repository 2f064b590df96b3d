"""AGC block kernel time vs tiles per block / warm-up constants on a C2-size capture (run on a GPU box)."""
import importlib, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    pdt = importlib.import_module("project-desert-tortoise_amd")
    iq = pdt.synth_capture(0, 50000, 600.0, seed=1234)
    d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True)
    d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
    print(f"TPB={os.environ.get('PDT_AGC_TPB')} K={os.environ.get('PDT_AGC_K')}: blocks {s.agc_blocks} fixes {s.agc_seam_fixes} "
          f"agc_block {kt['agc_block'][1]:.3f} agc_fix {kt['agc_fix'][1]:.3f} fir {kt['fir'][1]:.3f} gpu_ms {s.gpu_ms:.2f}", flush=True)
else:
    for tpb, K in (("2", "11"), ("1", "11"), ("4", "11"), ("1", "5"), ("2", "0.5"), ("1", "0.5")):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PDT_AGC_TPB=tpb, PDT_AGC_K=K))
