"""Debug aid: replay one case of tests/tools/fuzz.py (same draws), demodulate it in one shot and in pushes of BLOCK samples, and
compare the symbols each push adds with the one-shot symbol stream (first difference per push).
usage: python tools/probes/stream_debug.py <n_cases> <seed> <case> <block>"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
n_cases, seed, only, block = (int(x) for x in sys.argv[1:5])
rng = np.random.default_rng(seed)
for case in range(n_cases):
    argos = rng.random() < 0.25
    if argos:
        fs = 32000; secs = float(rng.uniform(2.0, 12.0)); chunk = int(rng.choice([0, 0, 1000, 2401, 4800, 777])); f0 = float(rng.uniform(-200, 200))
        kind, mode = 1, pdt.MODE_ARGOS
    else:
        fs = int(rng.choice([50000, 50000, 50000, 48000, 250000, 100000, 32000, 18750, 62500]))
        secs = float(rng.uniform(0.3, 8.0)) * (50000 / fs if fs > 50000 else 1.0)
        chunk = int(rng.choice([0, 0, 0, 1000, 3333, 10000, 25000, 260, 4096, int(rng.integers(300, 30000))]))
        f0 = float(rng.uniform(-4000, 4000)); kind, mode = 0, pdt.MODE_POES
    p = pdt.synth_params(kind, fs, f0, int(rng.integers(1, 1 << 30)))
    p.noise_gain = int(p.noise_gain * float(rng.choice([1, 1, 1, 2, 4, 7])))
    n = int(round(secs * fs))
    short = rng.random() < 0.15
    cut = int(rng.integers(0, min(n, 3 * (chunk or 10000)))) if short else None
    sampler = 1 if rng.random() < 0.15 else 0
    kw = {}
    if rng.random() < 0.3:
        kw = dict(pll_block=int(rng.integers(64, 6000)), pll_warm=int(rng.integers(0, 20000)),
                  agc_block=int(rng.integers(64, 12000)), agc_warm=int(rng.integers(0, 40000)))
    if rng.random() < 0.2:
        kw["gardner_band_pad"] = float(rng.choice([1 / 512, 1 / 64, 0.5]))
    rng.choice([2400, 5000, 12345, 777, 300, 1554])
    if case != only:
        continue
    iq = np.zeros((n, 2), dtype="<i2")
    pdt.synth_lib().pdt_synth_fill(C.byref(p), 0, n, iq.ctypes.data)
    if cut is not None:
        iq = iq[:cut]
    if os.environ.get("NO_GEOM"):
        kw = {}
    print("case", case, "argos" if argos else "poes", fs, len(iq), "chunk", chunk, kw)
    d = pdt.Demodulator(mode, fs, chunk=chunk, sampler=sampler, **kw)
    d.demod(iq)
    sym, bits, agc = d.stage(pdt.ST_SYM), d.stage(pdt.ST_BITS), d.stage(pdt.ST_AGC)
    want = d.frames_array()
    print("one-shot: symbols", len(sym), "bits", len(bits), "frames", [int(f["bit_index"]) for f in want])
    bs = "".join(chr(c) if c in (48, 49) else str(int(c)) for c in bits)
    pat = "0001011110000" if argos else "1110110111100010000"
    hits = [i + len(pat) - 1 for i in range(len(bs) - len(pat) + 1) if bs.startswith(pat, i)]
    print("sync matches end at bits", hits)
    for f in want:
        b = int(f["bit_index"]); print("  frame", b, bs[max(0, b - 40):b + 1], "|", bs[b + 1:b + 20])
    d.stream_begin()
    ps, pb, got = 0, 0, []
    for i in range(0, len(iq) + 1, block):
        fr = d.stream_push(iq[i:i + block]) if i < len(iq) else d.stream_end()
        got.append(fr)
        s = d.stats()
        ns, nb = int(s.symbols) - ps, int(s.bits) - pb
        if ns > 0:
            pad = 2 + (ps & 1)
            loc = d.stage(pdt.ST_SYM)
            assert len(loc) == pad + ns, (len(loc), pad, ns)
            ref = sym[ps:ps + ns]
            bad = np.nonzero(loc[pad:].view(np.uint64 if loc.itemsize == 8 else np.uint32) != ref.view(np.uint64 if ref.itemsize == 8 else np.uint32))[0]
            hist_ok = ps < 2 or (loc[pad - 2:pad].tobytes() == sym[ps - 2:ps].tobytes())
            if len(bad) or not hist_ok:
                print(f"push at {i}: symbols [{ps}, {ps + ns}) differ at local {bad[:5]} of {ns} (history ok {hist_ok})")
        if nb > 0:
            lb = d.stage(pdt.ST_BITS)
            kept = len(lb) - nb
            refb = bits[pb:pb + nb]
            badb = np.nonzero(lb[kept:] != refb)[0]
            kept_ok = lb[:kept].tobytes() == bits[pb - kept:pb].tobytes() if pb >= kept else None
            if 2440 <= pb <= 2540:
                print(f"push at {i}: bits [{pb}, {pb + nb}) kept {kept} local", "".join(chr(c) for c in lb[:kept]), "+", "".join(chr(c) for c in lb[kept:]), "frames", int(s.frames))
            if len(badb) or kept_ok is False:
                print(f"push at {i}: bits [{pb}, {pb + nb}) differ at local {badb[:8]} of {nb}; kept {kept} ok {kept_ok}")
        if len(fr):
            print(f"push at {i}: frames", [int(f["bit_index"]) for f in fr], "bits so far", int(s.bits))
        ps, pb = int(s.symbols), int(s.bits)
    got = np.concatenate(got)
    print("streamed identical", got.tobytes() == want.tobytes(), "symbols", ps, "bits", pb)
    d.close()
