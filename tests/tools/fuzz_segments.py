"""Randomised check of the overlapped file path (pdt_demod_file on a file large enough to be demodulated in segments while it is
read): random sample rates at INTERP 1 and above, chunk sizes, lengths, carrier offsets, noise, a noise-only lead (the lock then
happens in a later segment), random splits into 2 - 6 segments, table rows of 1 / 13 / 16 / other chunks -- the output FILE's bytes
must be the oracle's text, in half of the cases with the per-chunk reports on (pdt_keep_quality + pdt_set_progress: averagePhase
and the symbol / bit counts of every chunk against the oracle's chunk loop, every chunk handed on once and in order) -- and every
segment boundary lies on the grid where the segments take the whole-capture kernels
(k_mix_fir, k_agc_block_tr, rows of several chunks).  Usage: python tests/tools/fuzz_segments.py [n_cases] [seed]"""
import ctypes as C
import importlib, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pdt = importlib.import_module("project-desert-tortoise_amd")
from oracle import binding as orc
from math import gcd


def lcm(a, b):
    return a // gcd(a, b) * b


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t_start = time.time()
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, prefix="pdt_fuzzseg_")
for case in range(n_cases):
    fs = int(rng.choice([250000, 250000, 200000, 160000, 150000, 100000, 62500, 50000]))
    interp = int(round(150000.0 / fs))
    chunk = int(rng.choice([10000, 10000, 5000, 2600, 2400, 2000, 1300, 1000]))
    grid = lcm(lcm(chunk * 208, 64 * 26 * interp), 416)
    units = int(rng.integers(8, 14))
    n = grid * units + int(rng.integers(0, grid))            # (any length: the last segment ends where the capture does)
    if n > 40_000_000:
        n = grid * 8 + int(rng.integers(0, chunk * 3))
    f0 = float(rng.uniform(-4000, 4000))
    p = pdt.synth_params(0, fs, f0, int(rng.integers(1, 1 << 30)))
    p.noise_gain = int(p.noise_gain * float(rng.choice([1, 1, 2, 4, 6])))
    if rng.random() < 0.3:
        p.signal_start = int(n * float(rng.uniform(0.05, 0.7)))     # noise first: the PLL locks in a later segment
    iq = np.zeros((n, 2), dtype="<i2")
    pdt.synth_lib().pdt_synth_fill(C.byref(p), 0, n, iq.ctypes.data)
    k = int(rng.integers(2, 7))
    w = rng.uniform(0.3, 1.0, size=k)
    w = w / w.sum()
    span = int(rng.choice([0, 13, 16, 16, 8, 4, 26]))
    env = {"PDT_OVERLAP_MIN_MB": "1", "PDT_OVERLAP_SPLIT": ",".join(f"{x:.4f}" for x in w)}
    if span:
        env["PDT_GSPAN"] = str(span)
    if rng.random() < 0.15:
        env["PDT_SEG_PLAIN"] = "1"
    quality = rng.random() < 0.5
    o = orc.Oracle(orc.POES, fs, iq, chunk=chunk)
    wav = os.path.join(tmp, "c.wav")
    outp = os.path.join(tmp, "o.txt")
    pdt.write_wav(wav, fs, iq)
    os.environ.update(env)
    try:
        with pdt.Demodulator(pdt.MODE_POES, fs, chunk=chunk, profile=True).keep_pll(False) as d:
            calls = []
            if quality:
                d.keep_quality().set_progress(lambda first, r, st: calls.append((first, r)))
            fd = os.open(wav, os.O_RDONLY)
            fo = os.open(outp, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
            nb = d.demod_file_text(fd, 44, n, fo, 0)
            os.close(fd)
            os.close(fo)
            data = open(outp, "rb").read()
            s = d.stats()
            kt = d.kernel_times()
            ok = data == o.text() and nb == len(data) and data == d.text() and s.samples == n
            if quality:
                rep = d.chunk_reports()
                nc = (n + chunk - 1) // chunk
                avg = o.stage(orc.ST_AVG)
                cnt = o.stage(orc.ST_COUNTS).reshape(-1, 3)
                okq = len(rep) == nc and rep["avg_phase"].astype(avg.dtype).tobytes() == avg[:nc].tobytes()
                okq = okq and np.array_equal(rep["samples"], cnt[:nc, 0]) and np.array_equal(rep["symbols"], cnt[:nc, 1]) and \
                    np.array_equal(rep["bits"], cnt[:nc, 2]) and rep["frames"].sum() == s.frames
                okq = okq and len(calls) >= 2 and [c[0] for c in calls] == list(np.cumsum([0] + [len(c[1]) for c in calls[:-1]])) and \
                    np.concatenate([c[1] for c in calls]).tobytes() == rep.tobytes()
                if not okq:
                    print("     reports differ")
                ok = ok and okq
    finally:
        for key in env:
            os.environ.pop(key, None)
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} case {case}: fs {fs} chunk {chunk} n {n} ({n / grid:.2f} grid units) f0 {f0:.0f} noise x{p.noise_gain} lead {p.signal_start} "
          f"split {env['PDT_OVERLAP_SPLIT']} span {span} plain {'PDT_SEG_PLAIN' in env} frames {s.frames} lock {s.lock_sample} "
          f"last segment: {'mix_fir' if 'mix_fir' in kt else 'fir'} par {s.gardner_parallel} reports {int(quality)}", flush=True)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
print(f"{n_cases - bad}/{n_cases} identical in {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
