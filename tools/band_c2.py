import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
iq = pdt.synth_capture(0, 50000, 600.0, seed=1234)
for pad in (0.5, 0.25, 0.125):
    d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True, gardner_band_pad=pad)
    d.demod(iq); d.demod(iq); s = d.stats(); kt = d.kernel_times()
    print(f"pad {pad}: cand {s.gardner_candidates} ({s.gardner_candidates/2999:.0f}/chunk) walked {s.gardner_walked} full {s.gardner_full_domain} table {kt['gardner_table'][1]:.2f} ms chain {kt['gardner_chain'][1]:.2f} gpu_ms {s.gpu_ms:.2f}")
    d.close()
