"""How many distinct exits does a row of the sampler's boundary-state tables have?  ("have the span walker write the
symbols of the key the chain later selects when a row's surviving exits are <= 4 ... or a counter file showing the rows'
exit-count histogram that rules it out".)  Demodulates a capture resident in HBM and prints the histogram of the rows' distinct
exit keys (pdt_dev_span_rows).    usage: python tools/span_hist.py [config=c3] [seconds]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
pdt = importlib.import_module("project-desert-tortoise_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
c = bench.CONFIGS[cfg]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else c["seconds"]
fs = c["fs"]
n = int(round(secs * fs))
par = bench.capture_params(pdt, cfg, 1234, secs)
d_iq = bench.make_capture(pdt, par, n, min(32, os.cpu_count() or 8), device=torch.device("cuda", 0))
with pdt.Demodulator(pdt.MODE_POES, fs).keep_pll(False) as d:
    d.demod_device(d_iq.data_ptr(), n)
    L = pdt.lib()
    L.pdt_dev_span_rows.restype = C.c_uint64
    L.pdt_dev_span_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    rows = int(L.pdt_dev_span_rows(d._h, None, 0))
    out = np.zeros(max(rows, 1), dtype=np.uint32)
    L.pdt_dev_span_rows(d._h, out.ctypes.data, rows)
    st = d.stats()
ex = out[:rows]
ok = ex[ex != 0xFFFFFFFF]
print(f"{cfg}: {n} samples, {rows} table rows (build {pdt.build_tag()}), {int((ex == 0xFFFFFFFF).sum())} left untabulated; symbols {st.symbols}")
if len(ok):
    print(f"distinct exits of a row's first chunk: min {ok.min()}, median {int(np.median(ok))}, mean {ok.mean():.1f}, 95 % {int(np.percentile(ok, 95))}, max {ok.max()}")
    edges = [1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 1 << 30]
    for lo, hi in zip(edges[:-1], edges[1:]):
        k = int(((ok >= lo) & (ok < hi)).sum())
        print(f"  {lo:>4d} .. {hi - 1 if hi < (1 << 30) else 'more':>4}: {k:>7d} rows ({100.0 * k / len(ok):5.1f} %)")
    print(f"rows with at most 4 distinct exits: {100.0 * float((ok <= 4).mean()):.2f} %")
