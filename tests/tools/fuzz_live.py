"""Randomised GPU-vs-oracle comparison of the two sound-card twins' chains (PDT_CHAIN_LIVE): the ARGOS twin (the float build of
the ARGOS chain, ARGOSdemodPortAudio) and the POES twin (POESTIPdemodPortAudio) -- sample rates, block sizes (for ARGOS every
residue of the allocator's slack behind a float block), float32 / PCM16 input, a spectrum-inverted receiver, noise levels, very
short and empty captures, block geometries; every stage bit-identical, streamed frames = one-shot frames.
Usage: python tests/tools/fuzz_live.py [n_cases] [seed]"""
import ctypes as C
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
from oracle import binding as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t_start = time.time()
for case in range(n_cases):
    argos = rng.random() < 0.65
    fs = int(rng.choice([48000, 48000, 32000, 44100]))
    if argos:
        secs = float(rng.uniform(1.0, 14.0))
        chunk = int(rng.choice([0, 2400, 2401, 2402, 2403, 1000, 777, 4801, int(rng.integers(300, 6000))]))
        f0 = float(rng.uniform(-250, 250))
        kind, mode, omode = 1, pdt.MODE_ARGOS, orc.ARGOS
    else:
        secs = float(rng.uniform(0.5, 8.0))
        chunk = int(rng.choice([0, 2400, 1000, 10000, int(rng.integers(300, 12000))]))
        f0 = float(rng.uniform(-3500, 3500))
        kind, mode, omode = 0, pdt.MODE_POES, orc.POES
    p = pdt.synth_params(kind, fs, f0, int(rng.integers(1, 1 << 30)))
    p.noise_gain = int(p.noise_gain * float(rng.choice([1, 1, 2, 4, 7])))
    n = int(round(secs * fs))
    iq = np.zeros((n, 2), dtype="<i2")
    pdt.synth_lib().pdt_synth_fill(C.byref(p), 0, n, iq.ctypes.data)
    if rng.random() < 0.15:
        iq = iq[: int(rng.integers(0, min(n, 3 * (chunk or 2400))))]
    if rng.random() < 0.3:
        iq = iq.copy()
        iq[:, 1] = -iq[:, 1]                                   # spectrum-inverted receiver: inverse sync words
    as_f32 = rng.random() < 0.5
    src = iq.astype(np.float32) / np.float32(32768.0) * np.float32(rng.choice([1.0, 1.0, 0.05, 3.0])) if as_f32 else iq
    kw = {}
    if rng.random() < 0.3:
        kw = dict(pll_block=int(rng.integers(64, 6000)), pll_warm=int(rng.integers(0, 20000)),
                  agc_block=int(rng.integers(64, 12000)), agc_warm=int(rng.integers(0, 40000)))
    o = orc.Oracle(omode, fs, src, chunk=chunk or 2400, chain=1)
    d = pdt.Demodulator(mode, fs, chunk=chunk, chain=pdt.CHAIN_LIVE, **kw)
    (d.demod_raw if as_f32 else d.demod)(src)
    ok = d.text() == o.text()
    for sg, so in ((pdt.ST_PLL, orc.ST_PLL), (pdt.ST_LOCK, orc.ST_LOCK), (pdt.ST_FIR, orc.ST_FIR), (pdt.ST_AGC, orc.ST_AGC),
                   (pdt.ST_SYM, orc.ST_SYM), (pdt.ST_SYMIDX, orc.ST_SYMIDX), (pdt.ST_BITS, orc.ST_BITS)):
        a, b = d.stage(sg), o.stage(so)
        ok = ok and len(a) == len(b) and a.tobytes() == np.asarray(b, dtype=a.dtype).tobytes()
    want = d.frames_array().tobytes()
    d.stream_begin()
    blk = int(rng.choice([2400, 2400, 5000, 777, 300, 12345]))
    parts = [d.stream_push(src[i:i + blk]) for i in range(0, len(src), blk)] + [d.stream_end()]
    ok = ok and np.concatenate(parts).tobytes() == want and d.text() == o.text()
    s = d.stats()
    d.close()
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} case {case}: {'argos' if argos else 'poes'} twin fs {fs} n {len(iq)} chunk {chunk} f0 {f0:.0f} noise x{p.noise_gain} "
          f"{'f32' if as_f32 else 'pcm'} block {blk} geom {kw} frames {s.frames}", flush=True)
print(f"{n_cases - bad}/{n_cases} identical in {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
