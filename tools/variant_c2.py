"""Time the C2 workload with every library variant in variants/ (tile-parameter experiments; run on a GPU box)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import importlib
    sys.path.insert(0, ROOT)
    pdt = importlib.import_module("project-desert-tortoise_amd")
    secs = float(os.environ.get("PDT_SECS", "600"))
    iq = pdt.synth_capture(0, 50000, secs, seed=1234)
    d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True)
    d.demod(iq); d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
    print(os.path.basename(os.environ.get("PDT_LIBPDT_PATH", "default")), f"gpu_ms {s.gpu_ms:.2f} walked {s.gardner_walked} pllfix {s.pll_seam_fixes} agcfix {s.agc_seam_fixes} frames {s.frames} | " +
          " ".join(f"{k} {v[1]:.2f}" for k, v in kt.items()), flush=True)
else:
    for lib in sorted(glob.glob(os.path.join(ROOT, "variants", "*.so"))):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PDT_LIBPDT_PATH=lib))
