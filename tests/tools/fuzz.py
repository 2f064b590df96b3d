"""Randomised GPU-vs-oracle comparison: sample rates, chunk sizes, capture lengths, carrier offsets, noise levels,
block geometries, samplers.  Every stage must be bit-identical.  Usage: python tests/tools/fuzz.py [n_cases] [seed]"""
import ctypes as C
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
from oracle import binding as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # replay the draws, run just this case and say which stage differs
bad = 0
t_start = time.time()
for case in range(n_cases):
    argos = rng.random() < 0.25
    if argos:
        fs = 32000
        secs = float(rng.uniform(2.0, 12.0))
        chunk = int(rng.choice([0, 0, 1000, 2401, 4800, 777]))
        f0 = float(rng.uniform(-200, 200))
        kind, mode, omode = 1, pdt.MODE_ARGOS, orc.ARGOS
    else:
        fs = int(rng.choice([50000, 50000, 50000, 48000, 250000, 100000, 32000, 18750, 62500]))
        secs = float(rng.uniform(0.3, 8.0)) * (50000 / fs if fs > 50000 else 1.0)
        chunk = int(rng.choice([0, 0, 0, 1000, 3333, 10000, 25000, 260, 4096, int(rng.integers(300, 30000))]))
        f0 = float(rng.uniform(-4000, 4000))
        kind, mode, omode = 0, pdt.MODE_POES, orc.POES
    p = pdt.synth_params(kind, fs, f0, int(rng.integers(1, 1 << 30)))
    p.noise_gain = int(p.noise_gain * float(rng.choice([1, 1, 1, 2, 4, 7])))
    n = int(round(secs * fs))
    lead = float(rng.random())
    if not argos and lead < 0.2:
        p.signal_start = int(n * lead * 2.0)                   # noise in front of the signal (up to 40 % of the capture)
    iq = np.zeros((n, 2), dtype="<i2")
    pdt.synth_lib().pdt_synth_fill(C.byref(p), 0, n, iq.ctypes.data)
    if rng.random() < 0.15:
        iq = iq[: int(rng.integers(0, min(n, 3 * (chunk or 10000))))]          # very short / empty captures
    sampler = 1 if rng.random() < 0.15 else 0
    kw = {}
    if rng.random() < 0.3:
        kw = dict(pll_block=int(rng.integers(64, 6000)), pll_warm=int(rng.integers(0, 20000)),
                  agc_block=int(rng.integers(64, 12000)), agc_warm=int(rng.integers(0, 40000)))
    if rng.random() < 0.2:
        kw["gardner_band_pad"] = float(rng.choice([1 / 512, 1 / 64, 0.5]))
    span = int(rng.choice([0, 0, 0, 2, 3, 4, 8, 16]))         # table rows of several chunks (what hour-long captures take by default)
    if only >= 0 and case != only:
        rng.choice([2400, 5000, 12345, 777, 300, 1554])                              # (the streaming block size drawn further down)
        continue
    o = orc.Oracle(omode, fs, iq, chunk=chunk, sampler=sampler, math_mode=orc.MATH_LIBM)
    if span:
        os.environ["PDT_GSPAN"] = str(span)
    d = pdt.Demodulator(mode, fs, chunk=chunk, sampler=sampler, **kw)
    os.environ.pop("PDT_GSPAN", None)
    d.demod(iq)
    ok = d.text() == o.text()
    for sg, so in ((pdt.ST_PLL, orc.ST_PLL), (pdt.ST_FIR, orc.ST_FIR), (pdt.ST_AGC, orc.ST_AGC), (pdt.ST_SYM, orc.ST_SYM),
                   (pdt.ST_SYMIDX, orc.ST_SYMIDX), (pdt.ST_BITS, orc.ST_BITS)):
        a, b = d.stage(sg), o.stage(so)
        same = len(a) == len(b) and a.tobytes() == np.asarray(b, dtype=a.dtype).tobytes()
        if only >= 0 and not same:
            bb = np.asarray(b, dtype=a.dtype)
            m = min(len(a), len(bb))
            diff = np.nonzero(a[:m].view(np.uint8).reshape(m, -1) != bb[:m].view(np.uint8).reshape(m, -1))[0]
            print(f"  stage {sg}: lengths {len(a)} / {len(bb)}, {len(np.unique(diff))} elements differ, first at {diff[0] if len(diff) else None}")
        ok = ok and same
    # streaming must give the same frames as the one-shot call
    want = d.frames_array().tobytes()
    d.stream_begin()
    blk = int(rng.choice([2400, 5000, 12345, 777, 300, 1554]))
    parts = [d.stream_push(iq[i:i + blk]) for i in range(0, len(iq), blk)] + [d.stream_end()]
    if only >= 0:
        for b2 in [int(x) for x in os.environ.get("FUZZ_BLOCKS", "").split(",") if x]:
            d.stream_begin()
            pp = [d.stream_push(iq[i:i + b2]) for i in range(0, len(iq), b2)] + [d.stream_end()]
            got = np.concatenate(pp)
            print(f"  block {b2}: streamed frames identical {got.tobytes() == want}", [int(f["bit_index"]) for f in got])
        print("  text identical", d.text() == o.text(), "; streaming identical", np.concatenate(parts).tobytes() == want, "block", blk)
        if d.text() != o.text():
            print("  gpu text:\n" + d.text().decode(errors="replace") + "  oracle text:\n" + o.text().decode(errors="replace"))
            fr = np.frombuffer(want, dtype=pdt.FRAME_DTYPE)
            st = np.concatenate(parts)
            print("  one-shot frames (bit_index, nbytes, complete):", [(int(f["bit_index"]), int(f["nbytes"]), int(f["complete"])) for f in fr])
            print("  streamed frames:", [(int(f["bit_index"]), int(f["nbytes"]), int(f["complete"])) for f in st])
            print("  bits", d.stats().bits, "symbols", d.stats().symbols)
    ok = ok and np.concatenate(parts).tobytes() == want
    s = d.stats()
    d.close()
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} case {case}: {'argos' if argos else 'poes'} fs {fs} n {len(iq)} chunk {chunk} f0 {f0:.0f} noise x{p.noise_gain} "
          f"sampler {sampler} span {span} lead {p.signal_start} geom {kw} frames {s.frames} pllfix {s.pll_seam_fixes} par {s.gardner_parallel} walked {s.gardner_walked}", flush=True)
print(f"{n_cases - bad}/{n_cases} identical in {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
