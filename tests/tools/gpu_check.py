"""Quick stage-by-stage comparison of the HIP path with the oracle (run on a GPU box)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
from oracle import binding as orc


def compare(name, a, b):
    n = min(len(a), len(b))
    same = len(a) == len(b) and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    if same:
        print(f"  {name:8s} identical ({len(a)})")
        return True
    av, bv = a[:n], b[:n]
    neq = np.nonzero(av.view(np.uint8).reshape(n, -1) != bv.view(np.uint8).reshape(n, -1))[0]
    first = int(neq[0]) if len(neq) else n
    print(f"  {name:8s} DIFFERENT: len {len(a)} vs {len(b)}, first diff at {first}, ndiff {len(np.unique(neq))}")
    if first < n:
        print("     gpu", av[first:first + 4], "orc", bv[first:first + 4])
    return False


def run(label, mode, rate, iq, chunk=0, **kw):
    print(f"== {label}: {len(iq)} samples @ {rate} Hz chunk {chunk or 'default'}")
    t0 = time.time()
    o = orc.Oracle(mode, rate, iq, chunk=chunk, math_mode=orc.MATH_LIBM)
    t1 = time.time()
    d = pdt.Demodulator(mode, rate, chunk=chunk, profile=True, **kw)
    d.demod(iq)
    t2 = time.time()
    d.demod(iq)
    t3 = time.time()
    s = d.stats()
    print(f"  oracle {t1 - t0:.3f}s  gpu first {t2 - t1:.3f}s second {t3 - t2:.3f}s gpu_ms {s.gpu_ms:.3f}")
    print(f"  lock gpu {s.lock_sample} {s.lock_freq_hz:.2f}Hz norm {s.norm_factor:.7f} | orc {o.lock_sample} {o.lock_freq_hz:.2f}Hz norm {o.norm_factor:.7f}")
    print(f"  pll blocks {s.pll_blocks} fixes {s.pll_seam_fixes}; agc blocks {s.agc_blocks} fixes {s.agc_seam_fixes}; sym {s.symbols} bits {s.bits} frames {s.frames} gardner_parallel {s.gardner_parallel} walked {s.gardner_walked} fulldomain {s.gardner_full_domain} cand {s.gardner_candidates}")
    ok = True
    ok &= compare("pll", d.stage(pdt.ST_PLL), o.stage(orc.ST_PLL))
    if mode == pdt.MODE_ARGOS:
        ok &= compare("lock", d.stage(pdt.ST_LOCK), o.stage(orc.ST_LOCK))
    ok &= compare("fir", d.stage(pdt.ST_FIR), o.stage(orc.ST_FIR))
    ok &= compare("agc", d.stage(pdt.ST_AGC), o.stage(orc.ST_AGC))
    ok &= compare("sym", d.stage(pdt.ST_SYM), o.stage(orc.ST_SYM))
    ok &= compare("symidx", d.stage(pdt.ST_SYMIDX), o.stage(orc.ST_SYMIDX))
    ok &= compare("bits", d.stage(pdt.ST_BITS), o.stage(orc.ST_BITS))
    tg, to = d.text(), o.text()
    print("  text", "identical" if tg == to else f"DIFFERENT ({len(tg)} vs {len(to)})", len(tg))
    if tg != to:
        print(tg[:200]); print(to[:200])
    ok &= tg == to
    for k, (n, ms) in d.kernel_times().items():
        print(f"     {k:14s} {ms:9.3f} ms")
    return ok


if __name__ == "__main__":
    allok = True
    rate, iq = pdt.read_wav(os.path.join(ROOT, "tests/golden/5sec_clip.wav"))
    allok &= run("5sec_clip", pdt.MODE_POES, rate, iq)
    allok &= run("5sec_clip c=3333", pdt.MODE_POES, rate, iq, chunk=3333)
    allok &= run("5sec_clip small blocks", pdt.MODE_POES, rate, iq, pll_block=5000, pll_warm=12000, agc_block=20000, agc_warm=100000)
    for fs, secs in ((50000, 6.0), (250000, 6.0), (18750, 6.0), (50000, 60.0)):
        iq = pdt.synth_capture(0, fs, secs)
        allok &= run(f"synth {fs} {secs}s", pdt.MODE_POES, fs, iq)
    for seed, f0, secs, chunk in ((99, 120.0, 13.0, 0), (12, -90.0, 20.0, 1000), (13, 60.0, 30.0, 2401)):
        iq = pdt.synth_capture(1, 32000, secs, f0_hz=f0, seed=seed)
        allok &= run(f"argos seed {seed} chunk {chunk}", pdt.MODE_ARGOS, 32000, iq, chunk=chunk)
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
