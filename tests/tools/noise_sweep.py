"""Noisy captures: GPU vs oracle (every stage) while the noise of the synthetic capture is scaled up; also shows how the
repair / fallback paths are exercised (run on a GPU box)."""
import ctypes as C
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
from oracle import binding as orc


def capture(fs, secs, seed, noise_mult):
    p = pdt.synth_params(0, fs, 1000.0, seed)
    p.noise_gain = int(p.noise_gain * noise_mult)
    n = int(round(secs * fs))
    out = np.zeros((n, 2), dtype="<i2")
    pdt.synth_lib().pdt_synth_fill(C.byref(p), 0, n, out.ctypes.data)
    return out


ok = True
for mult in [float(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 8, 12]:
    iq = capture(50000, 60.0, 77, mult)
    o = orc.Oracle(orc.POES, 50000, iq)
    d = pdt.Demodulator(pdt.MODE_POES, 50000, profile=True)
    d.demod(iq); d.demod(iq)
    s = d.stats()
    same = d.text() == o.text()
    for st_g, st_o in ((pdt.ST_PLL, orc.ST_PLL), (pdt.ST_FIR, orc.ST_FIR), (pdt.ST_AGC, orc.ST_AGC), (pdt.ST_SYM, orc.ST_SYM),
                       (pdt.ST_SYMIDX, orc.ST_SYMIDX), (pdt.ST_BITS, orc.ST_BITS)):
        a, b = d.stage(st_g), o.stage(st_o)
        same = same and len(a) == len(b) and a.tobytes() == np.asarray(b, dtype=a.dtype).tobytes()
    sm, _ = d.tip_check()
    ok = ok and same
    print(f"noise x{mult:g}: identical {same}; lock {s.lock_sample} frames {s.frames} (error-free {sm['good_frames']}) gpu_ms {s.gpu_ms:.2f} "
          f"pll fixes {s.pll_seam_fixes}/{s.pll_blocks} agc fixes {s.agc_seam_fixes} gardner walked {s.gardner_walked} full {s.gardner_full_domain} "
          f"cand/chunk {s.gardner_candidates // max(1, s.samples // 10000)}", flush=True)
    d.close()
print("ALL OK" if ok else "MISMATCH")
