"""The C-ABI library: builds, loads, exports every symbol include/pdt.h declares, and fails
loudly when no GPU is present (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def header_functions(name="pdt.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pdt_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built(pdt):
    assert os.path.exists(pdt.LIBPDT_PATH), "run `make` (hipcc cross-compiles gfx950 without a GPU)"


def test_exports_every_declared_symbol(pdt):
    names = header_functions()
    assert len(names) >= 15
    L = C.CDLL(pdt.LIBPDT_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pdt.h but not exported by libpdt.so"
    assert set(pdt.ABI_SYMBOLS) == set(names)
    # the test-only entry (include/pdt_dev.h): the developer switches' registry
    dev = header_functions("pdt_dev.h")
    assert dev == sorted(pdt.DEV_SYMBOLS)
    for n in dev:
        assert hasattr(L, n)


def test_the_product_never_reads_the_environment(pdt):
    """Developer switches reach the library through pdt_dev_set only: no getenv in libpdt.so, the host
    programs or the gather library."""
    for rel in ("csrc/libpdt.so", "csrc/libpdtgather.so"):
        path = os.path.join(os.path.dirname(os.path.dirname(pdt.LIBPDT_PATH)), rel)
        out = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True).stdout
        assert "getenv" not in out, rel


def test_library_contains_gfx950_code_object(pdt):
    out = subprocess.run(["strings", "-a", pdt.LIBPDT_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert "k_gardner" in out and "k_pll_phase" in out and "k_fir_interp" in out


def test_struct_layouts_match_header(pdt):
    assert C.sizeof(pdt.Frame) == 136
    assert C.sizeof(pdt.Config) == 80
    assert pdt.FRAME_DTYPE.itemsize == 136
    assert pdt.TIP_DTYPE.itemsize == 12 and C.sizeof(pdt.TipSummary) == 56
    # the header itself must compile as plain C and agree on the sizes
    import tempfile
    src = ('#include "pdt.h"\n#include "pdt_gather.h"\nint main(void){return (sizeof(pdt_config)==80 && sizeof(pdt_frame)==136 && sizeof(pdt_tip_frame)==12 '
           '&& sizeof(pdt_tip_summary)==56 && sizeof(pdt_stats)>0) ? 0 : 1;}\n')
    with tempfile.TemporaryDirectory() as tmp:
        cfile = os.path.join(tmp, "abi.c")
        open(cfile, "w").write(src)
        exe = os.path.join(tmp, "abi")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe, cfile],
                       check=True)
        assert subprocess.run([exe]).returncode == 0


def test_no_gpu_fails_loudly(pdt, gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(pdt.PdtError, match="no usable HIP device"):
        pdt.Demodulator(pdt.MODE_POES, 50000)


def test_host_helpers_need_no_gpu(pdt, golden):
    for fs in golden["params"]["poes_rates"]:
        taps, interp = pdt.make_lpf(pdt.MODE_POES, fs)
        ref = np.fromfile(os.path.join(GOLDEN, f"taps_poes_{fs}.f32"), dtype=np.float32)
        assert interp == round(150000 / fs) and len(taps) == 26 * interp
        assert taps.tobytes() == ref.tobytes(), f"MakeLPFIR taps differ from the reference's at {fs} Hz"
    taps, interp = pdt.make_lpf(pdt.MODE_ARGOS, 32000)
    assert interp == 1
    assert taps.tobytes() == np.fromfile(os.path.join(GOLDEN, "taps_argos_32000.f64"), dtype=np.float64).tobytes()
    with pytest.raises(pdt.PdtError, match="interpolation factor"):
        pdt.make_lpf(pdt.MODE_POES, 400000)          # rint(150000/Fs) == 0 (reference divides by zero)


def test_wav_header_parse(pdt):
    L = pdt.lib()
    hdr = open(os.path.join(GOLDEN, "5sec_clip.wav"), "rb").read(44)
    vals = [C.c_uint32() for _ in range(5)]
    assert L.pdt_wav_parse_header(hdr, *[C.byref(v) for v in vals]) == 0
    rate, ch, bits, fmt, data = [v.value for v in vals]
    assert (rate, ch, bits, fmt, data) == (50000, 2, 16, 1, 250195 * 4)


def test_cli_binaries_fail_without_gpu(gpu_available):
    exe = os.path.join(ROOT, "bin", "demodPOES")
    if not os.path.exists(exe):
        pytest.skip("bin/demodPOES not built")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "No wave file specified" in r.stdout
    if not gpu_available:
        r = subprocess.run([exe, "-o", "/tmp/pdt_cli_test.txt", os.path.join(GOLDEN, "5sec_clip.wav")], capture_output=True,
                           text=True)
        assert r.returncode == 1 and "GPU demodulator unavailable" in r.stdout
        assert not os.path.exists("/tmp/pdt_cli_test.txt")


def test_hand_issued_lds_loads_are_the_only_users_of_m0(pdt, tmp_path):
    """The AGC and PLL walkers issue their look-ahead ring by hand (ring_issue in csrc/pdt_kernels_front.h): `s_mov_b32 m0` +
    `global_load_lds_dwordx4`, without telling the compiler that M0 is overwritten.  That is only sound while the compiler
    keeps no value of its own in M0 inside those kernels: check the shipped code object."""
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not installed")
    # the library is several translation units (round 6): its .hip_fatbin section holds one offload bundle per unit, back to
    # back -- every unit's gfx950 code object is looked at
    import struct
    fat = str(tmp_path / "fat.bin")
    subprocess.run([tools[0], "--dump-section", f".hip_fatbin={fat}", pdt.LIBPDT_PATH], check=True)
    blob = open(fat, "rb").read()
    magic, dis, units, at = b"__CLANG_OFFLOAD_BUNDLE__", "", 0, 0
    while (at := blob.find(magic, at)) >= 0:
        n = struct.unpack_from("<Q", blob, at + 24)[0]
        off = at + 32
        for _ in range(n):
            o, size, tlen = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24: off + 24 + tlen]
            off += 24 + tlen
            if b"gfx950" in triple and size:
                co = str(tmp_path / f"dev{units}.co")
                open(co, "wb").write(blob[at + o: at + o + size])
                dis += subprocess.run([tools[2], "-d", co], check=True, capture_output=True, text=True).stdout
                units += 1
        at += len(magic)
    assert units >= 4, f"{units} gfx950 code object(s) in libpdt.so: the chain's units are missing"
    kernel, per_kernel = None, {}
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel = m.group(1)
            continue
        if "global_load_lds" in line:
            per_kernel.setdefault(kernel, [0, 0])[0] += 1
        if re.search(r"\bm0\b", line):
            assert "s_mov_b32 m0," in line, f"{kernel}: unexpected use of m0: {line.strip()}"
            per_kernel.setdefault(kernel, [0, 0])[1] += 1
    assert per_kernel, "no hand-issued LDS loads found: is the ring still there?"
    for k, (loads, movs) in per_kernel.items():
        assert any(w in k for w in ("k_agc_", "k_pll_phase", "k_pll_head", "k_pll_tail", "k_pll_fix", "k_lock_ema")), f"m0 / LDS-direct load in an unexpected kernel: {k}"
        assert loads == movs and loads > 0, f"{k}: {loads} LDS-direct loads but {movs} writes of m0"


def test_format_records_equals_printf(pdt):
    """pdt_format_records prints "%.5f" without printf (exact decimal rounding of the binary value): compare with Python's
    printf-style formatting (== glibc's) on float32-derived stamps, arbitrary doubles, ties and extremes."""
    import numpy as np
    rng = np.random.default_rng(5)
    times = np.concatenate([
        rng.random(20000).astype(np.float32).astype(np.float64) * 600.0,           # float-built time stamps (POES)
        rng.random(20000) * 4000.0,                                                # double stamps (ARGOS)
        np.array([0.0, 0.5e-5, 1.5e-5, 2.5e-5, 0.000005, 0.000015, 0.125, 128.0, 512.0, 1024.0, 3600.0, 99999.999995,
                  1e-300, 5e-324, 1e14, 0.28009, 4.98649, 2.675, 1.0000049999999999, 123456.789015,
                  # beyond the integer path (printf fallback): up to 315 characters of time stamp must fit the line buffer
                  9.99999999999999e14, 1e15, 1e58, 1e300, 1.7976931348623157e308, np.inf, -1.0, np.nan]),
        (np.arange(0, 4000, dtype=np.float64) + 0.5) / 1e5,                        # decimal ties as doubles (mostly inexact)
        np.ldexp(np.arange(1, 2000, dtype=np.float64), -17),                        # exact binary fractions incl. true ties
    ])
    fr = np.zeros(len(times), dtype=pdt.FRAME_DTYPE)
    fr["time"] = times
    fr["nbytes"] = rng.integers(0, 105, len(times))
    fr["bytes"] = rng.integers(0, 256, (len(times), 104))
    fr["inverted"] = rng.integers(0, 2, len(times))
    fr["complete"] = rng.integers(0, 2, len(times))
    assert pdt.format_frames(fr) == pdt.format_frames_py(fr)
    assert pdt.format_frames(fr[:0]) == b""


def test_write_records_equals_format_records(pdt, tmp_path):
    """pdt_write_records (slices formatted and written side by side with pwrite) leaves exactly the bytes of
    pdt_format_records in the file, from the descriptor's position on, and the position behind them; a descriptor that
    cannot seek (a pipe) takes the text in order.  Host only."""
    import numpy as np
    L = pdt.lib()
    assert L.pdt_abi_version() == 4
    rng = np.random.default_rng(11)
    for n in (0, 1, 7, 2048, 2049, 40000):
        fr = np.zeros(n, dtype=pdt.FRAME_DTYPE)
        fr["time"] = rng.random(n).astype(np.float32).astype(np.float64) * 3600.0
        fr["nbytes"] = rng.integers(0, 105, n)
        fr["bytes"] = rng.integers(0, 256, (n, 104))
        fr["inverted"] = rng.integers(0, 2, n)
        fr["complete"] = rng.integers(0, 2, n)
        want = pdt.format_frames(fr)
        path = str(tmp_path / f"w{n}.txt")
        fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
        os.write(fd, b"head\n")
        nb = C.c_uint64(123)
        assert L.pdt_write_records(fr.ctypes.data, n, fd, C.byref(nb)) == 0
        assert nb.value == len(want)
        assert os.lseek(fd, 0, os.SEEK_CUR) == 5 + len(want)
        os.write(fd, b"tail")
        os.close(fd)
        assert open(path, "rb").read() == b"head\n" + want + b"tail"
    r, w = os.pipe()
    fr = fr[:300]
    want = pdt.format_frames(fr)
    import threading
    got = []
    th = threading.Thread(target=lambda: got.append(b"".join(iter(lambda: os.read(r, 1 << 16), b""))))
    th.start()
    assert L.pdt_write_records(fr.ctypes.data, len(fr), w, None) == 0
    os.close(w)
    th.join()
    os.close(r)
    assert got[0] == want
    assert L.pdt_write_records(None, 3, 1, None) == -1          # PDT_ERR_ARG


def test_gather_library_exports_its_entry_point(pdt):
    """include/pdt_gather.h: the RCCL gather of frame records lives in a library of its own (libpdt.so has no RCCL dependency)."""
    path = os.path.join(os.path.dirname(pdt.LIBPDT_PATH), "libpdtgather.so")
    assert os.path.exists(path), "run `make`"
    needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
    assert "librccl" in needed and "libpdt.so" in needed
    assert "librccl" not in subprocess.run(["readelf", "-d", pdt.LIBPDT_PATH], capture_output=True, text=True).stdout
    syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    for name in re.findall(r"\b(pdt_gather\w*)\s*\(", open(os.path.join(ROOT, "include", "pdt_gather.h")).read()):
        assert f" T {name}" in syms, f"{name} declared in include/pdt_gather.h but not exported by libpdtgather.so"
    assert " T pdt_gather_frames" in syms and " T pdt_gatherer_gather" in syms and " T pdt_gather_unpad" in syms


def test_compat_libraries_export_the_reference_prototypes(pdt):
    """libpdt_compat_{poes,argos}.so (host/pdt_compat.c): every live stage function of common/*.h and the per-program byte
    synchroniser is there under the reference's own name (no compute call without a GPU: symbols only)."""
    import ctypes
    csrc = os.path.dirname(pdt.LIBPDT_PATH)
    ctypes.CDLL(pdt.LIBPDT_PATH, mode=ctypes.RTLD_GLOBAL)
    common = ["StaticGain", "NormalizingAGC", "Squelch", "CarrierTrackPLL", "LowPassFilterInterp", "LowPassFilter", "MakeLPFIR",
              "GardenerClockRecovery", "MMClockRecovery", "sign", "ManchesterDecode"]
    for name, own in (("libpdt_compat_poes.so", "ByteSyncOnSyncword"), ("libpdt_compat_argos.so", "FindSyncWords")):
        path = os.path.join(csrc, name)
        assert os.path.exists(path), f"{name} not built (make)"
        lib = ctypes.CDLL(path)
        for sym in common + [own]:
            assert hasattr(lib, sym), f"{name}: {sym}"


def test_one_hip_runtime_whatever_the_import_order():
    """A PyTorch-ROCm wheel brings its own libamdhip64; loaded after the system copy libpdt.so links, it is a second runtime in
    the process and sees no GPU.  The binding loads torch's copy first when there is one (project-desert-tortoise_amd
    ._share_torch_hip_runtime), so the package may be imported before torch: one runtime is mapped either way."""
    code = (
        "import importlib, sys\n"
        "m = importlib.import_module('project-desert-tortoise_amd'); m.lib()\n"
        "import torch\n"
        "libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))\n"
        "print(len(libs), libs)\n"
        "sys.exit(0 if len(libs) == 1 else 1)\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
