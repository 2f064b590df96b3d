"""Probe (round 5): is there time to be had from running the chain's stages side by side?
The c3 step is a serial chain of kernels that each use part of the chip (PLL walkers: 588 of 1 024 SIMDs; mix + FIR and the
Gardner tables: vector-issue-bound; AGC: HBM-bound).  Pipelining the capture over time segments on alternating streams would make
stage s of segment k run beside stage s - 1 of segment k + 1.  Upper bound without writing that pipeline: TWO independent
half-length captures through two contexts (own streams) at once -- every stage of one may overlap any stage of the other -- against
the same two one after the other, and against one full-length capture; also with the second capture started a fixed delay after
the first (its PLL beside the first one's post-PLL stages).  Usage: python tools/probes/stage_concurrency.py"""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
pdt = importlib.import_module("project-desert-tortoise_amd")

fs = 250000
n_full, n_half = 900_000_000, 450_000_000
dev = torch.device("cuda", 0)
d_full = bench.make_capture(pdt, bench.capture_params(pdt, "c3", 1234), n_full, 32, device=dev)
d_b = bench.make_capture(pdt, bench.capture_params(pdt, "c3", 4321), n_half, 32, device=dev)
full = pdt.Demodulator(0, fs).keep_pll(False)
a = pdt.Demodulator(0, fs).keep_pll(False)
b = pdt.Demodulator(0, fs).keep_pll(False)
pa, pb = d_full.data_ptr(), d_b.data_ptr()          # A = the first half of the full capture


def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def both(delay_ms=0.0):
    def run_b():
        if delay_ms:
            time.sleep(delay_ms * 1e-3)
        b.demod_device(pb, n_half)
    th = threading.Thread(target=run_b)
    th.start()
    a.demod_device(pa, n_half)
    th.join()


for _ in range(3):
    full.demod_device(pa, n_full); a.demod_device(pa, n_half); b.demod_device(pb, n_half)
print("one capture of 900 M samples:            median %.2f ms (min %.2f)" % timed(lambda: full.demod_device(pa, n_full)))
print("one capture of 450 M samples:            median %.2f ms (min %.2f)" % timed(lambda: a.demod_device(pa, n_half)))
print("two of 450 M, one after the other:       median %.2f ms (min %.2f)" % timed(lambda: (a.demod_device(pa, n_half), b.demod_device(pb, n_half))))
for d in (0.0, 2.0, 4.0, 6.0, 8.0):
    print("two of 450 M at once, second %.0f ms later: median %.2f ms (min %.2f)" % ((d,) + timed(lambda: both(d))))
print("batched (one launch per stage for both): median %.2f ms (min %.2f)" % timed(lambda: pdt.demod_batch([a, b], [pa, pb], [n_half, n_half])))
