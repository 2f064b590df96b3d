#!/usr/bin/env python3
"""Round-4 probe: the c3 step in a process WITHOUT torch (the library on the system's HIP runtime, input from host memory, no
caller stream): are the stage times those of the benched process?  usage: python tools/probes/c3_notorch.py [torch]
(with `torch`: import torch first, as bench.py does, everything else the same)"""
import concurrent.futures as cf
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
with_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
if with_torch:
    import torch
    torch.cuda.init()
pdt = importlib.import_module("project-desert-tortoise_amd")
if not with_torch:
    pdt._share_torch_hip_runtime = lambda: None           # the system's libamdhip64, as the C programs get it
fs, n = 250000, 900_000_000
p = pdt.synth_params(0, fs, 1000.0)
S = pdt.synth_lib()
S.pdt_synth_sine_table()
iq = np.empty((n, 2), dtype="<i2")
piece = 1 << 21
t0 = time.time()
with cf.ThreadPoolExecutor(8) as ex:
    list(ex.map(lambda off: S.pdt_synth_fill(C.byref(p), off, min(piece, n - off), iq[off:off + piece].ctypes.data), range(0, n, piece)))
print(f"capture generated in {time.time() - t0:.1f} s", flush=True)
de = pdt.Demodulator(0, fs, chunk=10000, profile=True)
for r in range(3):
    t0 = time.time()
    de.demod(iq)
    st = de.stats()
    kt = de.kernel_times()
    print(("with torch imported first" if with_torch else "no torch in the process"), f"call {r}: {1e3 * (time.time() - t0):.0f} ms wall,",
          {k: round(v[1] / max(v[0], 1), 3) for k, v in kt.items() if k.startswith("pll") or k in ("mix_fir", "agc_block")}, flush=True)
maps = open("/proc/self/maps").read()
print("runtime:", sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l}))
