import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pdt = importlib.import_module("project-desert-tortoise_amd")
iq = pdt.synth_capture(0, 50000, 60.0, seed=1234)
d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True, pll_block=2500)
d.demod(iq)
s = d.stats()
print("fixes", s.pll_seam_fixes, d.kernel_times()["pll_fix"])
