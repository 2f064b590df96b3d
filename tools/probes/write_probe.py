import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib, numpy as np
pdt = importlib.import_module("project-desert-tortoise_amd")
L = pdt.lib()
n = 36000
rng = np.random.default_rng(1)
fr = np.zeros(n, dtype=pdt.FRAME_DTYPE)
fr["time"] = np.arange(n) * 0.1
fr["nbytes"] = 104
fr["bytes"] = rng.integers(0, 256, (n, 104))
fr["complete"] = 1
for rep in range(4):
    t0 = time.perf_counter()
    fd = os.open("/dev/shm/wtest.txt", os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    t1 = time.perf_counter()
    nb = C.c_uint64(0)
    L.pdt_write_records(fr.ctypes.data, n, fd, C.byref(nb))
    t2 = time.perf_counter()
    os.close(fd)
    t3 = time.perf_counter()
    buf = C.create_string_buffer(n * 352)
    t4 = time.perf_counter()
    k = L.pdt_format_records(fr.ctypes.data, n, buf, n * 352)
    t5 = time.perf_counter()
    print(f"open {1e3*(t1-t0):.2f} write_records {1e3*(t2-t1):.2f} close {1e3*(t3-t2):.2f}  | alloc {1e3*(t4-t3):.2f} format only {1e3*(t5-t4):.2f} bytes {nb.value}")
os.unlink("/dev/shm/wtest.txt")
