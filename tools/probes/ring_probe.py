"""Where the ring sampler's time goes: ARGOS bench capture vs an all-squelched capture of the same length,
ring kernel vs the old two-buffer kernel (PDT_GARDNER_NORING).  Run on the GPU box."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pdt = importlib.import_module("project-desert-tortoise_amd")

def run(iq, env):
    for k in ("PDT_GARDNER_NORING",):
        os.environ.pop(k, None)
    os.environ.update(env)
    t = torch.from_numpy(iq.reshape(-1).copy()).to("cuda:0")
    torch.cuda.synchronize()
    with pdt.Demodulator(pdt.MODE_ARGOS, 32000, profile=True) as d:
        for _ in range(3):
            d.demod_device(t.data_ptr(), len(iq))
        g = []
        for _ in range(5):
            d.demod_device(t.data_ptr(), len(iq))
            g.append(d.kernel_times()["gardner"][1])
        st = d.stats()
    return round(min(g), 3), "ms gardner; symbols", st.symbols, "frames", st.frames

sig = pdt.synth_capture(1, 32000, 300.0, f0_hz=300.0, seed=1234)
zero = np.zeros_like(sig)
for name, iq in (("signal", sig), ("zeros", zero)):
    for env in ({}, {"PDT_GARDNER_NORING": "1"}):
        print(name, env, run(iq, env))
