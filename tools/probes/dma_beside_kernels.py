"""Probe (round 5): how fast do pinned-host -> HBM copies run while the chain's kernels are running?
The overlapped ingest delivered ~8 GB/s while a segment's kernels ran and 52 GB/s while the GPU idled.  This measures the
copies alone, beside the resident c3 step looping on another stream, and beside a plain streaming kernel, for the runtime's
copy paths (HSA_ENABLE_SDMA=0 forces shader copies).  Usage: python tools/probes/dma_beside_kernels.py [seconds_of_capture]"""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
pdt = importlib.import_module("project-desert-tortoise_amd")

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1200.0
fs = 250000
n = int(secs * fs)
dev = torch.device("cuda", 0)
par = bench.capture_params(pdt, "c3", 1234)
d_iq = bench.make_capture(pdt, par, n, 16, device=dev)
dm = pdt.Demodulator(0, fs, device=0).keep_pll(False)
st_main = torch.cuda.Stream()
dm.set_stream(st_main.cuda_stream)
dm.demod_device(d_iq.data_ptr(), n)
torch.cuda.synchronize()

GB = 1 << 30
host = torch.empty(GB, dtype=torch.uint8).pin_memory()
dst = torch.empty(GB, dtype=torch.uint8, device=dev)
copy_streams = [torch.cuda.Stream() for _ in range(4)]


def copy_once(nstreams=4, piece=8 << 20):
    ev = []
    t0 = time.perf_counter()
    k = 0
    for off in range(0, GB, piece):
        s = copy_streams[k % nstreams]
        with torch.cuda.stream(s):
            dst[off:off + piece].copy_(host[off:off + piece], non_blocking=True)
        k += 1
    for s in copy_streams[:nstreams]:
        s.synchronize()
    return GB / (time.perf_counter() - t0) / 1e9


stop = False


def chain_loop():
    while not stop:
        dm.demod_device(d_iq.data_ptr(), n)


def stream_loop():
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(20):
                a.mul_(1.0001)
            s.synchronize()


print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"), " capture", n, "samples")
for ns in (1, 4):
    print(f"  copies alone, {ns} stream(s): " + ", ".join(f"{copy_once(ns):.1f}" for _ in range(3)) + " GB/s")
for name, fn in (("the c3-like chain looping", chain_loop), ("an elementwise streaming kernel looping", stream_loop)):
    stop = False
    th = threading.Thread(target=fn)
    th.start()
    time.sleep(0.3)
    for ns in (1, 4):
        print(f"  copies beside {name}, {ns} stream(s): " + ", ".join(f"{copy_once(ns):.1f}" for _ in range(3)) + " GB/s")
    stop = True
    th.join()
    torch.cuda.synchronize()
