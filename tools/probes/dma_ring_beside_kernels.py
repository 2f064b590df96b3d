"""Probe (round 5): the ingest's copy pipeline -- a ring of 16 pinned 8 MiB slots, one submitter that queues a copy when a slot
is free and frees a slot when its copy's event has completed -- without any file reading, alone and beside the chain's kernels,
over 1 / 2 / 4 copy streams.  Tells whether the overlapped ingest is held up by the copies (submission / completion beside
running kernels) or by the readers.  Usage: python tools/probes/dma_ring_beside_kernels.py"""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
pdt = importlib.import_module("project-desert-tortoise_amd")

fs = 250000
n = 300_000_000
dev = torch.device("cuda", 0)
par = bench.capture_params(pdt, "c3", 1234)
d_iq = bench.make_capture(pdt, par, n, 16, device=dev)
dm = pdt.Demodulator(0, fs, device=0).keep_pll(False)
st_main = torch.cuda.Stream()
dm.set_stream(st_main.cuda_stream)
dm.demod_device(d_iq.data_ptr(), n)
torch.cuda.synchronize()

PIECE = 8 << 20
NSLOT = 16
TOTAL = 3600 << 20
slots = [torch.empty(PIECE, dtype=torch.uint8).pin_memory() for _ in range(NSLOT)]
dst = torch.empty(TOTAL, dtype=torch.uint8, device=dev)
streams = [torch.cuda.Stream() for _ in range(4)]
events = [torch.cuda.Event() for _ in range(NSLOT)]


def ring(ns, depth=NSLOT):
    inflight = []          # (slot) in submission order
    free = list(range(depth))
    t0 = time.perf_counter()
    k = 0
    nsp = TOTAL // PIECE
    while k < nsp:
        while inflight and events[inflight[0]].query():
            free.append(inflight.pop(0))
        if not free:
            time.sleep(15e-6)
            continue
        s = free.pop(0)
        st = streams[k % ns]
        with torch.cuda.stream(st):
            dst[k * PIECE:(k + 1) * PIECE].copy_(slots[s], non_blocking=True)
            events[s].record(st)
        inflight.append(s)
        k += 1
    for st in streams:
        st.synchronize()
    return TOTAL / (time.perf_counter() - t0) / 1e9


stop = False


def chain_loop():
    while not stop:
        dm.demod_device(d_iq.data_ptr(), n)


print("ring of", NSLOT, "slots of", PIECE >> 20, "MiB,", TOTAL >> 20, "MiB per pass")
for ns in (1, 2, 4):
    print(f"  alone, {ns} stream(s): " + ", ".join(f"{ring(ns):.1f}" for _ in range(3)) + " GB/s")
for depth in (4, 8):
    print(f"  alone, 2 streams, {depth} slots: " + ", ".join(f"{ring(2, depth):.1f}" for _ in range(2)) + " GB/s")
th = threading.Thread(target=chain_loop)
th.start()
time.sleep(0.3)
for ns in (1, 2, 4):
    print(f"  beside the chain (300 M samples, looping), {ns} stream(s): " + ", ".join(f"{ring(ns):.1f}" for _ in range(3)) + " GB/s")
stop = True
th.join()
