"""k_pll_phase / k_pll_head time at the c3 geometry (250 ksps, an hour) against the tracking warm-up length: slope = ns per
step, intercept = guess + wide-band / acquisition-gain stages + block (run on a GPU box)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pdt = importlib.import_module("project-desert-tortoise_amd")
fs, seconds = 250000, float(os.environ.get("SECONDS_C3", "3600"))
n = int(fs * seconds)
dev = torch.device("cuda", 0)
par = bench.capture_params(pdt, "c3", 1234)
d_iq = bench.make_capture(pdt, par, n, 32, device=dev, fs=fs)
for warm in (0, 25000, 50000, 100000, 150000):
    d = pdt.Demodulator(pdt.MODE_POES, fs, profile=True, pll_warm=warm).keep_pll(False)
    for _ in range(3):
        d.demod_device(d_iq.data_ptr(), n)
    s = d.stats(); kt = d.kernel_times()
    print(f"warm {warm if warm else 'default'}: pll_phase {kt['pll_phase'][1]:.3f} ms  head {kt['pll_head'][1]:.3f}  fixes {s.pll_seam_fixes}  pll_fix {kt['pll_fix'][1]:.3f}  step {s.gpu_ms:.2f}", flush=True)
    d.close()
