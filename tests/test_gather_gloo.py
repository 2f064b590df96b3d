"""The N>1 path of bench.py: one capture per rank, frame records gathered on rank 0.
Runs on CPU with the gloo backend (world size 2); frame records come from the oracle here,
on the GPU box they come from libpdt."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    pdt = importlib.import_module("project-desert-tortoise_amd")
    from oracle import binding as orc
    sys.path.insert(0, os.path.join(ROOT))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fs = 50000
    iq = pdt.synth_capture(0, fs, 2.0 + rank, seed=100 + rank)      # ragged: different frame counts per rank
    o = orc.Oracle(orc.POES, fs, iq, keep_stages=False)
    frames = np.zeros(len(o.frames()), dtype=pdt.FRAME_DTYPE)
    for i, f in enumerate(o.frames()):
        frames[i]["time"] = f.time
        frames[i]["bit_index"] = f.bit_index
        frames[i]["inverted"] = f.inverted
        frames[i]["nbytes"] = f.nbytes
        frames[i]["complete"] = f.complete
        frames[i]["bytes"] = np.frombuffer(bytes(f.bytes), dtype=np.uint8)
    gathered = bench.gather_frames(frames, device=torch.device("cpu"))
    if rank == 0:
        q.put([(g.tobytes(), pdt.format_frames(g)) for g in gathered])
        q.put(o.text())
    else:
        q.put(o.text())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_frames_world2():
    ctx = mp.get_context("spawn")
    q0, q1 = ctx.Queue(), ctx.Queue()
    port = _free_port()
    p0 = ctx.Process(target=_worker, args=(0, 2, port, q0))
    p1 = ctx.Process(target=_worker, args=(1, 2, port, q1))
    p0.start(); p1.start()
    gathered = q0.get(timeout=120)
    text0 = q0.get(timeout=120)
    text1 = q1.get(timeout=120)
    p0.join(60); p1.join(60)
    assert p0.exitcode == 0 and p1.exitcode == 0
    assert len(gathered) == 2
    # rank 0 can write both captures' output files byte-exactly from the gathered records
    assert gathered[0][1] == text0
    assert gathered[1][1] == text1
    assert len(gathered[0][0]) != len(gathered[1][0])     # ragged counts survived the padded all_gather


RAGGED = [0, 5, 17, 0, 1, 300, 2, 64]          # frames per rank: zero counts, a lone frame, one rank far above the others


def _records(pdt, rank, n):
    rng = np.random.default_rng(1000 + rank)
    fr = np.zeros(n, dtype=pdt.FRAME_DTYPE)
    fr["time"] = rng.random(n) * 3600.0
    fr["bit_index"] = np.arange(n) * 832 + rank
    fr["nbytes"] = 104
    fr["complete"] = 1
    fr["bytes"] = rng.integers(0, 256, (n, 104))
    return fr


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    pdt = importlib.import_module("project-desert-tortoise_amd")
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for rnd in range(2):                                  # twice: every count 0 in the second round
        mine = _records(pdt, rank, RAGGED[rank] if rnd == 0 else 0)
        got = bench.gather_frames(mine, device=torch.device("cpu"))
        if rank == 0:
            q.put([g.tobytes() for g in got])
    dist.barrier()
    dist.destroy_process_group()


def test_gather_frames_world8_ragged_and_empty():
    """configs[4]'s shape: 8 ranks.  Ragged counts with zeros among them, then a round in which nobody has a frame; the padded
    exchange goes through libpdtgather's pdt_gather_plan / pdt_gather_unpad, the functions demodMulti's RCCL gather uses."""
    import importlib
    sys.path.insert(0, ROOT)
    pdt = importlib.import_module("project-desert-tortoise_amd")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    first = q.get(timeout=180)
    second = q.get(timeout=180)
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    assert [len(b) for b in first] == [c * pdt.FRAME_DTYPE.itemsize for c in RAGGED]
    for r in range(8):
        assert first[r] == _records(pdt, r, RAGGED[r]).tobytes()
    assert second == [b""] * 8


def test_gather_plan_and_unpad_host_only():
    import ctypes as C
    sys.path.insert(0, ROOT)
    import bench
    L = bench.gather_lib()
    counts = np.array([3, 0, 1], dtype=np.uint64)
    nmax, off = C.c_uint64(0), np.zeros(4, dtype=np.uint64)
    assert L.pdt_gather_plan(counts.ctypes.data, 3, C.byref(nmax), off.ctypes.data) == 0
    assert nmax.value == 3 and list(off) == [0, 3, 3, 4]
    padded = np.arange(3 * 3 * 8, dtype=np.uint8)
    out = np.zeros(4 * 8, dtype=np.uint8)
    assert L.pdt_gather_unpad(padded.ctypes.data, counts.ctypes.data, 3, 3, 8, out.ctypes.data) == 0
    assert list(out) == list(range(24)) + list(range(48, 56))
    zero = np.zeros(2, dtype=np.uint64)
    assert L.pdt_gather_plan(zero.ctypes.data, 2, C.byref(nmax), None) == 0 and nmax.value == 1
    assert L.pdt_gather_unpad(padded.ctypes.data, np.array([4], dtype=np.uint64).ctypes.data, 1, 3, 8, out.ctypes.data) != 0
