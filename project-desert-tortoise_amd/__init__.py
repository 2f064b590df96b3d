"""project-desert-tortoise_amd -- MI355X (gfx950) POES-TIP / ARGOS IQ demodulation chain.

Python is only a thin ctypes veneer over ``csrc/libpdt.so`` (hand-written HIP kernels behind the
C ABI of ``include/pdt.h``) and ``synth/libpdtsynth.so`` (integer-only synthetic captures).
There is **no CPU fallback**: if the HIP library is missing, or no GPU is visible when a
context is opened, the call fails loudly.

The directory name contains hyphens, so import it with::

    import importlib; pdt = importlib.import_module("project-desert-tortoise_amd")

Reference behaviour mirrored here (file:line in nebarnix/Project-Desert-Tortoise):
  * ``Demodulator`` == one run of the chunk loop POESTIPdemod/main.c:373-492 or
    ARGOSdemod/main.c:250-306 over a whole capture.
  * ``Demodulator.text()`` == the bytes POESTIPdemod/ByteSync.c:62-69,96-101 /
    ARGOSdemod/ByteSync.c:62-70,99-103 write to minorFrames_*.txt / packets_*.txt.
"""
from __future__ import annotations

import ctypes as C
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPDT_PATH = os.environ.get("PDT_LIBPDT_PATH") or os.path.join(_HERE, "csrc", "libpdt.so")   # override: tuning experiments only
LIBSYNTH_PATH = os.path.join(_HERE, "synth", "libpdtsynth.so")

MODE_POES, MODE_ARGOS = 0, 1
SAMPLER_GARDNER, SAMPLER_MM = 0, 1
CHAIN_FILE, CHAIN_LIVE = 0, 1          # CHAIN_LIVE: the sound-card twin's constants and stage order (POES)
ST_PLL, ST_LOCK, ST_FIR, ST_AGC, ST_SYM, ST_SYMIDX, ST_BITS, ST_BITSYM, ST_AGC_RAW = range(9)


class PdtError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("sample_rate", C.c_uint32),
        ("chunk", C.c_uint64),
        ("norm_override", C.c_double),
        ("device", C.c_int32),
        ("profile", C.c_int32),
        ("pll_block", C.c_uint32),
        ("pll_warm", C.c_uint32),
        ("agc_block", C.c_uint32),
        ("agc_warm", C.c_uint32),
        ("gardner_band_pad", C.c_double),
        ("sampler", C.c_int32),
        ("chain", C.c_int32),
        ("mm_step_range", C.c_double),
        ("mm_kp", C.c_double),
    ]


class LoopParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("pll_freq_range_hz", "pll_lock_threshold", "pll_lock_alpha", "pll_loopbw_acq", "pll_loopbw_track",
                                          "agc_attack", "agc_decay", "gardner_baud", "gardner_step_range", "gardner_kp", "manchester_threshold")] \
               + [("zero_mask", C.c_uint32), ("reserved_", C.c_uint32)]


class Frame(C.Structure):
    _fields_ = [
        ("time", C.c_double),
        ("bit_index", C.c_int64),
        ("time_src", C.c_int64),
        ("inverted", C.c_uint8),
        ("nbytes", C.c_uint8),
        ("complete", C.c_uint8),
        ("pad", C.c_uint8),
        ("bytes", C.c_uint8 * 104),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("out_samples", C.c_uint64),
        ("symbols", C.c_uint64),
        ("bits", C.c_uint64),
        ("frames", C.c_uint64),
        ("lock_sample", C.c_int64),
        ("lock_freq_hz", C.c_double),
        ("norm_factor", C.c_double),
        ("avg_phase", C.c_double),
        ("interp", C.c_uint32),
        ("ntaps", C.c_uint32),
        ("pll_blocks", C.c_uint32),
        ("pll_seam_fixes", C.c_uint32),
        ("agc_blocks", C.c_uint32),
        ("agc_seam_fixes", C.c_uint32),
        ("gpu_ms", C.c_double),
        ("gardner_parallel", C.c_uint32),
        ("gardner_walked", C.c_uint32),
        ("gardner_full_domain", C.c_uint32),
        ("sync_overflow", C.c_uint32),
        ("gardner_candidates", C.c_uint64),
        ("ingest_ms", C.c_double),
        ("alloc_ms", C.c_double),
        ("segments", C.c_uint32),
        ("windowed", C.c_uint32),
        ("ingest_direct", C.c_uint32),
        ("ingest_numa_node", C.c_int32),
    ]


class TipSummary(C.Structure):
    _fields_ = [("frames_checked", C.c_uint64), ("good_frames", C.c_uint64), ("good_chunks", C.c_uint64),
                ("bad_chunks", C.c_uint64), ("spacecraft", C.c_int32), ("day", C.c_int32), ("t0_ms", C.c_int64),
                ("time_frames", C.c_uint64)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p)     # == pdt_progress_fn
# == pdt_chunk_report (48 bytes): what the reference's chunk loop knows after every chunk
CHUNK_DTYPE = np.dtype([("samples", "<u8"), ("avg_phase", "<f8"), ("symbols", "<u8"), ("bits", "<u8"), ("frames", "<u8"),
                        ("time0", "<f8")])

TIP_DTYPE = np.dtype([("minor_id", "<u2"), ("spacecraft", "u1"), ("parity", "u1"), ("checked", "u1"), ("has_time", "u1"),
                      ("day", "<u2"), ("day_ms", "<i4")])        # == pdt_tip_frame (12 bytes)


class ManchesterState(C.Structure):
    """pdt_manchester_state: ManchesterDecode's statics (a zeroed record = before the first call)"""
    _fields_ = [("current", C.c_double), ("previous", C.c_double), ("clockmod", C.c_uint32), ("even_odd", C.c_uint32)]


class PllState(C.Structure):
    """pdt_pll_state: CarrierTrackPLL's statics (a zeroed record = before the first call)"""
    _fields_ = [("started", C.c_int32), ("locked", C.c_int32), ("lock_index", C.c_int64), ("lock_freq_hz", C.c_double),
                ("phase", C.c_double), ("freq", C.c_double), ("avg_phase", C.c_double), ("locksig", C.c_double),
                ("sweep", C.c_double)]


class GardnerState(C.Structure):
    """pdt_gardner_state: GardenerClockRecovery's statics (a zeroed record = before the first call)"""
    _fields_ = [("next_sample", C.c_double), ("prev_bit", C.c_double), ("half_sample", C.c_double)]


class MmState(C.Structure):
    """pdt_mm_state: MMClockRecovery's statics (a zeroed record = before the first call)"""
    _fields_ = [("started", C.c_int32), ("pad", C.c_int32), ("next_sample", C.c_double), ("step_size", C.c_double),
                ("sample_last", C.c_double)]


class AgcState(C.Structure):
    """pdt_agc_state: NormalizingAGC's static gain (a zeroed record = before the first call)"""
    _fields_ = [("started", C.c_int32), ("pad", C.c_int32), ("gain", C.c_double)]


class FirState(C.Structure):
    """pdt_fir_state: the low-pass filter's ring (a zeroed record = before the first call)"""
    _fields_ = [("count", C.c_uint64), ("history", C.c_double * 64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint32), ("total_ms", C.c_double)]


# every symbol include/pdt.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "pdt_abi_version", "pdt_build_tag", "pdt_strerror", "pdt_device_count", "pdt_open", "pdt_close", "pdt_set_stream",
    "pdt_demod_pcm16", "pdt_demod_device", "pdt_demod_f32", "pdt_demod_device_f32", "pdt_demod_batch_device", "pdt_num_frames", "pdt_frames", "pdt_get_stats",
    "pdt_format_frames", "pdt_read_stage", "pdt_stage_len", "pdt_kernel_times", "pdt_make_lpf",
    "pdt_wav_parse_header", "pdt_time_axis", "pdt_stage_bytesync", "pdt_tip_check", "pdt_tip_frames",
    "pdt_stream_begin", "pdt_stream_push_pcm16", "pdt_stream_push_f32", "pdt_stream_end", "pdt_stream_frames",
    "pdt_keep_quality", "pdt_chunk_reports", "pdt_stage_manchester", "pdt_stage_fir", "pdt_stage_agc", "pdt_stage_squelch", "pdt_stage_pll", "pdt_stage_gardner", "pdt_stage_static_gain", "pdt_stage_mm",
    "pdt_keep_presquelch", "pdt_keep_pll", "pdt_stage_bytesync_from", "pdt_demod_fd", "pdt_format_records", "pdt_stream_retained", "pdt_host_math", "pdt_get_device",
    "pdt_write_frames", "pdt_write_records", "pdt_demod_file", "pdt_set_loop_params", "pdt_set_progress",
]
DEV_SYMBOLS = ["pdt_dev_set", "pdt_dev_span_rows"]        # include/pdt_dev.h (test-only)

_lib = None


def _share_torch_hip_runtime():
    """A PyTorch-ROCm wheel carries its own copy of the HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).
    Loaded after libpdt.so has brought in the system copy, it is a second runtime in the process and finds no GPU ("No HIP
    GPUs are available"); loaded first, libpdt.so's dependency resolves to it and both use one runtime -- the order `import
    torch` before this package always gave.  Make the order irrelevant: when a torch with a bundled runtime is installed
    (found without importing it), load that copy before libpdt.so.  Without torch nothing happens and libpdt.so uses the
    system runtime, as the C programs do."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libpdt.so (built in-tree by ``make`` / ``__graft_entry__.build()``); fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPDT_PATH):
        raise PdtError(
            f"{LIBPDT_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
            "There is no CPU implementation to fall back to."
        )
    _share_torch_hip_runtime()
    L = C.CDLL(LIBPDT_PATH)
    L.pdt_abi_version.restype = C.c_int
    L.pdt_build_tag.restype = C.c_char_p
    L.pdt_strerror.restype = C.c_char_p
    L.pdt_strerror.argtypes = [C.c_int]
    L.pdt_device_count.restype = C.c_int
    L.pdt_open.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.pdt_close.argtypes = [C.c_void_p]
    L.pdt_close.restype = None
    L.pdt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.pdt_demod_pcm16.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pdt_demod_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pdt_demod_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pdt_demod_device_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pdt_demod_fd.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int]
    L.pdt_format_records.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
    L.pdt_format_records.restype = C.c_uint64
    L.pdt_num_frames.argtypes = [C.c_void_p]
    L.pdt_num_frames.restype = C.c_uint64
    L.pdt_frames.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_uint64]
    L.pdt_frames.restype = C.c_uint64
    L.pdt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.pdt_format_frames.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    L.pdt_format_frames.restype = C.c_uint64
    L.pdt_read_stage.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
    L.pdt_read_stage.restype = C.c_int64
    L.pdt_stage_len.argtypes = [C.c_void_p, C.c_int]
    L.pdt_stage_len.restype = C.c_uint64
    L.pdt_kernel_times.argtypes = [C.c_void_p, C.POINTER(KernelTime), C.c_int]
    L.pdt_make_lpf.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pdt_wav_parse_header.argtypes = [C.c_char_p] + [C.POINTER(C.c_uint32)] * 5
    L.pdt_time_axis.argtypes = [C.c_int, C.c_uint32, C.c_uint64]
    L.pdt_time_axis.restype = C.c_double
    L.pdt_stage_bytesync.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pdt_demod_batch_device.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int]
    L.pdt_stream_begin.argtypes = [C.c_void_p]
    L.pdt_stream_push_pcm16.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.pdt_stream_push_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.pdt_stream_end.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.pdt_stream_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pdt_stream_frames.restype = C.c_uint64
    L.pdt_host_math.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.pdt_stream_retained.argtypes = [C.c_void_p]
    L.pdt_stream_retained.restype = C.c_uint64
    L.pdt_stage_manchester.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pdt_stage_manchester.restype = C.c_int
    L.pdt_stage_pll.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pdt_stage_pll.restype = C.c_int
    L.pdt_stage_gardner.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]
    L.pdt_stage_gardner.restype = C.c_int
    L.pdt_stage_static_gain.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.c_void_p]
    L.pdt_stage_static_gain.restype = C.c_int
    L.pdt_stage_mm.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pdt_stage_mm.restype = C.c_int
    L.pdt_stage_agc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_void_p]
    L.pdt_stage_agc.restype = C.c_int
    L.pdt_stage_squelch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_double]
    L.pdt_stage_squelch.restype = C.c_int
    L.pdt_stage_fir.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.pdt_stage_fir.restype = C.c_int
    L.pdt_keep_quality.argtypes = [C.c_void_p, C.c_int]
    L.pdt_keep_quality.restype = C.c_int
    L.pdt_keep_pll.argtypes = [C.c_void_p, C.c_int]
    L.pdt_keep_pll.restype = C.c_int
    L.pdt_chunk_reports.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pdt_chunk_reports.restype = C.c_uint64
    L.pdt_set_progress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pdt_tip_check.argtypes = [C.c_void_p, C.POINTER(TipSummary)]
    L.pdt_tip_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pdt_tip_frames.restype = C.c_uint64
    L.pdt_write_frames.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.pdt_write_records.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    L.pdt_dev_set.argtypes = [C.c_char_p, C.c_char_p]
    L.pdt_set_loop_params.argtypes = [C.c_void_p, C.POINTER(LoopParams)]
    L.pdt_demod_file.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    if L.pdt_abi_version() != 4:
        raise PdtError("libpdt.so ABI version mismatch")
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != 0:
        raise PdtError(f"{what}: {lib().pdt_strerror(rc).decode()} ({rc})")


def build_tag() -> str:
    """pdt_build_tag: which sources the loaded libpdt.so was built from (profiles/ files carry the same tag)"""
    return lib().pdt_build_tag().decode()


def make_lpf(mode: int, sample_rate: int) -> tuple[np.ndarray, int]:
    """MakeLPFIR as the reference's mains call it (LowPassFilter.c:127-175). Returns (taps, interp)."""
    L = lib()
    nt, ip = C.c_int(), C.c_int()
    _check(L.pdt_make_lpf(mode, sample_rate, None, C.byref(nt), C.byref(ip)), "pdt_make_lpf")
    taps = np.zeros(nt.value, dtype=np.float64 if mode == MODE_ARGOS else np.float32)
    _check(L.pdt_make_lpf(mode, sample_rate, taps.ctypes.data, None, None), "pdt_make_lpf")
    return taps, ip.value


def host_math(fn: int, x: np.ndarray):
    """pdt_host_math: the library's own sincos / sin / cos / sincosf / hypot / hypotf evaluated on the host (test hook)."""
    a = np.ascontiguousarray(x, dtype=np.float64)
    n = len(a) // 2 if fn in (4, 5) else len(a)
    o0, o1 = np.zeros(n), np.zeros(n)
    _check(lib().pdt_host_math(fn, a.ctypes.data, n, o0.ctypes.data, o1.ctypes.data), "pdt_host_math")
    return o0, o1


def time_axis(mode: int, sample_rate: int, m: int) -> float:
    return lib().pdt_time_axis(mode, sample_rate, m)


def read_wav(path: str) -> tuple[int, np.ndarray]:
    """44-byte canonical header only, like ReadWavHeader (wave.c:303-378); every byte after it is
    sample data (the reference reads to EOF, POESTIPdemod/main.c:373).  Returns (rate, int16[n,2])."""
    with open(path, "rb") as f:
        hdr = f.read(44)
        data = f.read()
    if len(hdr) < 44:
        raise PdtError("short WAV header")
    fmt, ch, rate = struct.unpack_from("<HHI", hdr, 20)
    bits = struct.unpack_from("<H", hdr, 34)[0]
    if fmt != 1 or ch != 2 or bits != 16:
        raise PdtError("need 16-bit PCM with 2 channels (I,Q)")
    n = len(data) // 4
    return rate, np.frombuffer(data, dtype="<i2", count=2 * n).reshape(n, 2)


def sync_dev_switches(L=None):
    """Mirror the process's PDT_* environment variables into the library's developer-switch registry (``pdt_dev_set``,
    include/pdt_dev.h: a TEST-ONLY entry -- the library itself never reads the environment, and the C host programs never
    call this).  Called before every ``pdt_open`` of this binding, so tests and sweeps keep saying
    ``os.environ["PDT_GSPAN"] = "4"`` around the opening of a context."""
    L = L or lib()
    L.pdt_dev_set(None, None)
    for k, v in os.environ.items():
        if k.startswith("PDT_"):
            L.pdt_dev_set(k.encode(), v.encode())


class Demodulator:
    """One capture -> minor frames / packets on one GPU (context of include/pdt.h)."""

    def __init__(self, mode: int, sample_rate: int, chunk: int = 0, norm_override: float = 0.0, device: int = 0,
                 profile: bool = False, pll_block: int = 0, pll_warm: int = 0, agc_block: int = 0, agc_warm: int = 0,
                 gardner_band_pad: float = 0.0, sampler: int = 0, mm_step_range: float = 0.0, mm_kp: float = 0.0, chain: int = 0):
        self._L = lib()
        self.mode = mode
        self.sample_rate = sample_rate
        cfg = Config(mode, sample_rate, chunk, norm_override, device, int(profile), pll_block, pll_warm, agc_block,
                     agc_warm, gardner_band_pad, sampler, chain, mm_step_range, mm_kp)
        self._h = C.c_void_p()
        sync_dev_switches(self._L)
        _check(self._L.pdt_open(C.byref(cfg), C.byref(self._h)), "pdt_open")
        self.chain = chain
        self.dtype = np.float64 if (mode == MODE_ARGOS and not chain) else np.float32      # (the ARGOS twin is the float build)

    def close(self):
        if self._h:
            self._L.pdt_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_loop_params(self, **kw):
        """``pdt_set_loop_params``: loop constants other than the mains' (fields of ``LoopParams``; 0 / absent = default)."""
        lp = LoopParams(**kw)
        # a constant given as 0 where 0 is a meaningful value (lock threshold, timing gain / clip, resync threshold) IS zero
        zero_bits = {"pll_lock_threshold": 1, "gardner_kp": 2, "gardner_step_range": 4, "manchester_threshold": 8}
        if "zero_mask" not in kw:
            lp.zero_mask = sum(b for k, b in zero_bits.items() if k in kw and kw[k] == 0)
        _check(self._L.pdt_set_loop_params(self._h, C.byref(lp)), "pdt_set_loop_params")
        return self

    def keep_presquelch(self, enable: bool = True):
        """Also keep the AGC output before Squelch (stage ST_AGC_RAW): what the reference's -r option dumps."""
        _check(self._L.pdt_keep_presquelch(self._h, int(enable)), "pdt_keep_presquelch")
        return self

    def keep_pll(self, enable: bool = True):
        """Whether the PLL output stream (ST_PLL) is written out where mix and filter run as one kernel (on by default)."""
        _check(self._L.pdt_keep_pll(self._h, int(enable)), "pdt_keep_pll")
        return self

    def keep_quality(self, enable: bool = True):
        """Also keep the per-chunk reports (CarrierTrackPLL's return value = averagePhase, symbol / bit / frame counts):
        what the reference's progress line shows (POESTIPdemod/main.c:457-481)."""
        _check(self._L.pdt_keep_quality(self._h, int(enable)), "pdt_keep_quality")
        return self

    def set_progress(self, fn=None):
        """``pdt_set_progress``: ``fn(first_chunk, reports, stats_so_far)`` is called with the reports (CHUNK_DTYPE array) of the
        chunks that have become final -- once per completed segment of an overlapped ``demod_file`` / ``demod_file_text``
        (from a thread of the library), once at the end of any other whole-capture call.  Needs ``keep_quality``."""
        if fn is None:
            self._progress_cb = None
            _check(self._L.pdt_set_progress(self._h, None, None), "pdt_set_progress")
            return self

        def tramp(_user, first, reports, n, st):
            arr = np.frombuffer(C.string_at(reports, int(n) * CHUNK_DTYPE.itemsize), dtype=CHUNK_DTYPE).copy()
            fn(int(first), arr, Stats.from_buffer_copy(C.string_at(st, C.sizeof(Stats))))

        self._progress_cb = PROGRESS_FN(tramp)                       # (kept alive as long as the context uses it)
        _check(self._L.pdt_set_progress(self._h, C.cast(self._progress_cb, C.c_void_p), None), "pdt_set_progress")
        return self

    def chunk_reports(self) -> np.ndarray:
        n = int(self._L.pdt_chunk_reports(self._h, None, 0))
        out = np.zeros(n, dtype=CHUNK_DTYPE)
        if n:
            got = int(self._L.pdt_chunk_reports(self._h, out.ctypes.data, n))
            out = out[:got]
        return out

    def set_stream(self, stream_handle: int):
        _check(self._L.pdt_set_stream(self._h, C.c_void_p(stream_handle)), "pdt_set_stream")

    def demod(self, iq: np.ndarray):
        """iq: int16 array of shape (n, 2) or flat interleaved I,Q in host memory."""
        a = np.ascontiguousarray(iq, dtype="<i2").reshape(-1)
        _check(self._L.pdt_demod_pcm16(self._h, a.ctypes.data, a.size // 2), "pdt_demod_pcm16")
        return self

    def demod_raw(self, iq: np.ndarray):
        """RAW capture: float32 array of shape (n, 2) or flat interleaved I,Q, used without normalisation."""
        a = np.ascontiguousarray(iq, dtype="<f4").reshape(-1)
        _check(self._L.pdt_demod_f32(self._h, a.ctypes.data, a.size // 2), "pdt_demod_f32")
        return self

    def demod_file(self, fd: int, byte_offset: int, nframes: int, fmt: int = 0):
        """The capture straight from an open file (descriptor `fd`): nframes I,Q pairs from byte_offset (44 after the
        canonical WAV header, 0 for RAW); fmt 0 = int16 pairs, 1 = float32 pairs."""
        _check(self._L.pdt_demod_fd(self._h, fd, byte_offset, nframes, fmt), "pdt_demod_fd")
        return self

    def demod_device(self, dev_ptr: int, nframes: int):
        """Input already resident in HBM (e.g. ``tensor.data_ptr()`` of an int16 torch tensor)."""
        _check(self._L.pdt_demod_device(self._h, C.c_void_p(dev_ptr), nframes), "pdt_demod_device")
        return self

    def bytesync(self, bits: np.ndarray):
        """Stage-level entry: only the sync-word search / frame extraction on a uint8 '0'/'1' array."""
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        _check(self._L.pdt_stage_bytesync(self._h, b.ctypes.data, b.size), "pdt_stage_bytesync")
        return self

    def _dt(self):
        return np.dtype("<f8") if (self.mode == MODE_ARGOS and not self.chain) else np.dtype("<f4")

    def stage_manchester(self, symbols: np.ndarray, resync_threshold: float, state: "ManchesterState | None" = None):
        """ManchesterDecode on these symbols alone (statics in `state`, updated in place): (bits '0'/'1', symbol index per bit)"""
        a = np.ascontiguousarray(symbols, dtype=self._dt())
        bits = np.zeros(a.size + 4, dtype=np.uint8)           # a resynchronisation can make consecutive symbols both end a bit
        bsym = np.zeros(a.size + 4, dtype=np.uint32)
        nb = C.c_uint64(0)
        _check(self._L.pdt_stage_manchester(self._h, a.ctypes.data, a.size, float(resync_threshold),
                                            C.addressof(state) if state is not None else None, bits.ctypes.data, bsym.ctypes.data,
                                            C.addressof(nb)), "pdt_stage_manchester")
        return bits[:nb.value], bsym[:nb.value]

    def stage_fir(self, x: np.ndarray, state: "FirState | None" = None) -> np.ndarray:
        """LowPassFilterInterp (POES) / LowPassFilter (ARGOS) on these inputs alone (ring in `state`, updated in place)"""
        a = np.ascontiguousarray(x, dtype=self._dt())
        out = np.zeros(a.size * self.stats_interp(), dtype=self._dt())
        _check(self._L.pdt_stage_fir(self._h, a.ctypes.data, a.size, C.addressof(state) if state is not None else None,
                                     out.ctypes.data), "pdt_stage_fir")
        return out

    def stage_pll(self, iq: np.ndarray, state: "PllState | None" = None):
        """CarrierTrackPLL on these samples alone (float32[n,2] = `float complex`, or int16[n,2] as the WAV holds them; statics in
        `state`, updated in place): (realDataOut, lockSignalStreamOut, return value)"""
        f32 = np.asarray(iq).dtype.kind == "f"
        a = np.ascontiguousarray(iq, dtype="<f4" if f32 else "<i2").reshape(-1)
        n = a.size // 2
        out = np.zeros(n, dtype=self._dt())
        lock = np.zeros(n, dtype=self._dt())
        ret = C.c_double(0)
        _check(self._L.pdt_stage_pll(self._h, a.ctypes.data, n, 1 if f32 else 0, C.addressof(state) if state is not None else None,
                                     out.ctypes.data, lock.ctypes.data, C.addressof(ret)), "pdt_stage_pll")
        return out, lock, ret.value

    def stage_gardner(self, buf: np.ndarray, n: int, state: "GardnerState | None" = None, neighbour: "np.ndarray | None" = None):
        """GardenerClockRecovery on buf[:n]; `buf` is the caller's whole buffer (what lies behind n is read as it stands);
        returns (symbols, pick index per symbol)"""
        a = np.ascontiguousarray(buf, dtype=self._dt())
        nb = np.ascontiguousarray(neighbour, dtype=self._dt()) if neighbour is not None else None
        cap = int(n / 4) + 64
        sym = np.zeros(cap, dtype=self._dt())
        pick = np.zeros(cap, dtype=np.uint64)
        ns = C.c_uint64(0)
        _check(self._L.pdt_stage_gardner(self._h, a.ctypes.data, int(n), a.size, nb.ctypes.data if nb is not None else None,
                                         C.addressof(state) if state is not None else None, sym.ctypes.data, pick.ctypes.data,
                                         C.addressof(ns)), "pdt_stage_gardner")
        return sym[:ns.value], pick[:ns.value]

    def stage_static_gain(self, iq: np.ndarray, level: float = 1.0) -> float:
        """StaticGain over these samples (int16[n,2] as the WAV holds them, or float32[n,2])"""
        f32 = np.asarray(iq).dtype.kind == "f"
        a = np.ascontiguousarray(iq, dtype="<f4" if f32 else "<i2").reshape(-1)
        g = C.c_double(0)
        _check(self._L.pdt_stage_static_gain(self._h, a.ctypes.data, a.size // 2, 1 if f32 else 0, float(level), C.addressof(g)),
               "pdt_stage_static_gain")
        return g.value

    def stage_mm(self, x: np.ndarray, state: "MmState | None" = None):
        """MMClockRecovery on these samples alone (statics in `state`, updated in place): (symbols, pick index per symbol)"""
        a = np.ascontiguousarray(x, dtype=self._dt())
        cap = int(a.size / 3) + 64
        sym = np.zeros(cap, dtype=self._dt())
        pick = np.zeros(cap, dtype=np.uint64)
        ns = C.c_uint64(0)
        _check(self._L.pdt_stage_mm(self._h, a.ctypes.data, a.size, C.addressof(state) if state is not None else None,
                                    sym.ctypes.data, pick.ctypes.data, C.addressof(ns)), "pdt_stage_mm")
        return sym[:ns.value], pick[:ns.value]

    def stage_agc(self, x: np.ndarray, initial: float, state: "AgcState | None" = None, attack: float = 0.0, decay: float = 0.0):
        """NormalizingAGC on these samples alone (gain in `state`, updated in place); returns the output (the reference works in place)"""
        a = np.array(x, dtype=self._dt(), copy=True)
        _check(self._L.pdt_stage_agc(self._h, a.ctypes.data, a.size, float(initial), float(attack), float(decay),
                                     C.addressof(state) if state is not None else None), "pdt_stage_agc")
        return a

    def stage_squelch(self, x: np.ndarray, lock: np.ndarray, threshold: float) -> np.ndarray:
        a = np.array(x, dtype=self._dt(), copy=True)
        l = np.ascontiguousarray(lock, dtype=self._dt())
        assert a.size == l.size
        _check(self._L.pdt_stage_squelch(self._h, a.ctypes.data, l.ctypes.data, a.size, float(threshold)), "pdt_stage_squelch")
        return a

    def stats_interp(self) -> int:
        taps, interp = make_lpf(self.mode, self.sample_rate)
        return int(interp)

    def frames(self) -> list[Frame]:
        n = self._L.pdt_num_frames(self._h)
        arr = (Frame * max(n, 1))()
        got = self._L.pdt_frames(self._h, arr, n)
        return [arr[i] for i in range(got)]

    def frames_array(self) -> np.ndarray:
        """Frames as a structured numpy array (same layout as ``pdt_frame``)."""
        n = self._L.pdt_num_frames(self._h)
        buf = np.zeros(n, dtype=FRAME_DTYPE)
        if n:
            self._L.pdt_frames(self._h, C.cast(buf.ctypes.data, C.POINTER(Frame)), n)
        return buf

    def stats(self) -> Stats:
        s = Stats()
        _check(self._L.pdt_get_stats(self._h, C.byref(s)), "pdt_get_stats")
        return s

    def text(self) -> bytes:
        cap = int(self._L.pdt_num_frames(self._h)) * 352 + 64          # a line is at most 24 + 3 * 104 + 1 characters
        buf = C.create_string_buffer(cap)
        n = self._L.pdt_format_frames(self._h, buf, cap)
        return buf.raw[:n]

    def demod_file_text(self, fd: int, byte_offset: int, nframes: int, text_fd: int, fmt: int = 0) -> int:
        """``pdt_demod_file``: capture file in, frame text out to ``text_fd`` in one call (a large POES file: in overlapped
        segments, each segment's text written while the next one runs); returns the number of text bytes written."""
        nb = C.c_uint64(0)
        _check(self._L.pdt_demod_file(self._h, int(fd), byte_offset, nframes, fmt, int(text_fd), C.byref(nb)), "pdt_demod_file")
        return int(nb.value)

    def write_frames(self, fd: int) -> int:
        """The text of the last run written to an open file descriptor (``pdt_write_frames``: formatted and written in slices
        by a few threads); returns the number of bytes."""
        nb = C.c_uint64(0)
        _check(self._L.pdt_write_frames(self._h, int(fd), C.byref(nb)), "pdt_write_frames")
        return int(nb.value)

    def stage_len(self, st: int) -> int:
        return int(self._L.pdt_stage_len(self._h, st))

    def stage(self, st: int, first: int = 0, count: int | None = None) -> np.ndarray:
        total = self._L.pdt_stage_len(self._h, st)
        if count is None:
            count = max(total - first, 0)
        dt = {ST_SYMIDX: np.int64, ST_BITS: np.uint8, ST_BITSYM: np.uint32}.get(st, self.dtype)
        out = np.zeros(count, dtype=dt)
        if count:
            got = self._L.pdt_read_stage(self._h, st, first, count, out.ctypes.data)
            if got < 0:
                _check(int(got), "pdt_read_stage")
            out = out[:got]
        return out

    # ---- streaming front end: push blocks, collect frames as they become final
    def stream_begin(self):
        _check(self._L.pdt_stream_begin(self._h), "pdt_stream_begin")
        return self

    def _stream_new(self, n: int) -> np.ndarray:
        buf = np.zeros(n, dtype=FRAME_DTYPE)
        if n:
            self._L.pdt_stream_frames(self._h, buf.ctypes.data, n)
        return buf

    def stream_push(self, iq: np.ndarray) -> np.ndarray:
        """Append a block (int16 or float32 I,Q pairs); returns the frames that became final with it."""
        n = C.c_uint64(0)
        if np.asarray(iq).dtype.kind == "f":
            a = np.ascontiguousarray(iq, dtype="<f4").reshape(-1)
            _check(self._L.pdt_stream_push_f32(self._h, a.ctypes.data, a.size // 2, C.byref(n)), "pdt_stream_push_f32")
        else:
            a = np.ascontiguousarray(iq, dtype="<i2").reshape(-1)
            _check(self._L.pdt_stream_push_pcm16(self._h, a.ctypes.data, a.size // 2, C.byref(n)), "pdt_stream_push_pcm16")
        return self._stream_new(n.value)

    def stream_end(self) -> np.ndarray:
        """The capture is over: returns the remaining frames; text()/stats()/frames_array() then describe all of it."""
        n = C.c_uint64(0)
        _check(self._L.pdt_stream_end(self._h, C.byref(n)), "pdt_stream_end")
        return self._stream_new(n.value)

    def stream_retained(self) -> int:
        """Input samples the device currently holds for the stream (history + not yet demodulated): bounded."""
        return int(self._L.pdt_stream_retained(self._h))

    def tip_check(self):
        """Frame validation of the last demodulation (the reference's MATLAB checkParity.m / daytimeDecode.m,
        on the GPU): (summary dict, per-frame structured array)."""
        sm = TipSummary()
        _check(self._L.pdt_tip_check(self._h, C.byref(sm)), "pdt_tip_check")
        n = self._L.pdt_num_frames(self._h)
        rec = np.zeros(n, dtype=TIP_DTYPE)
        if n:
            self._L.pdt_tip_frames(self._h, rec.ctypes.data, n)
        return {k: int(getattr(sm, k)) for k, _ in TipSummary._fields_}, rec

    def kernel_times(self) -> dict[str, tuple[int, float]]:
        n = self._L.pdt_kernel_times(self._h, None, 0)
        arr = (KernelTime * max(n, 1))()
        self._L.pdt_kernel_times(self._h, arr, n)
        return {arr[i].name.decode(): (arr[i].launches, arr[i].total_ms) for i in range(n)}


def demod_batch(demods, dev_ptrs, nframes):
    """Batched many-capture mode: demods[i] demodulates the int16 I/Q capture at device address dev_ptrs[i]
    (nframes[i] pairs); the kernels of all captures overlap on the GPU.  Results are read from each Demodulator."""
    n = len(demods)
    if not (n == len(dev_ptrs) == len(nframes)):
        raise ValueError("demod_batch: lists of different lengths")
    hs = (C.c_void_p * max(n, 1))(*[d._h for d in demods])
    ps = (C.c_void_p * max(n, 1))(*[C.c_void_p(int(p)) for p in dev_ptrs])
    ns = (C.c_uint64 * max(n, 1))(*[int(v) for v in nframes])
    _check(lib().pdt_demod_batch_device(hs, ps, ns, n), "pdt_demod_batch_device")


FRAME_DTYPE = np.dtype([
    ("time", "<f8"), ("bit_index", "<i8"), ("time_src", "<i8"), ("inverted", "u1"), ("nbytes", "u1"),
    ("complete", "u1"), ("pad", "u1"), ("bytes", "u1", (104,)), ("_tail", "u1", (4,)),   # C struct is padded to 136
])
assert FRAME_DTYPE.itemsize == C.sizeof(Frame)


def format_frames(frames: np.ndarray) -> bytes:
    """Text of a (possibly gathered) frame array: ``pdt_format_records`` (host-only C, no GPU needed)."""
    a = np.ascontiguousarray(frames, dtype=FRAME_DTYPE)
    L = lib()
    n = L.pdt_format_records(a.ctypes.data, len(a), None, 0)
    buf = C.create_string_buffer(n + 1)
    L.pdt_format_records(a.ctypes.data, len(a), buf, n)
    return buf.raw[:n]


def format_frames_py(frames: np.ndarray) -> bytes:
    """The same text through Python's own "%.5f" / "%.2X" (the reference's printf formats); used by the tests."""
    out = []
    for f in frames:
        out.append((b"%.5fi " if f["inverted"] else b"%.5f ") % f["time"])
        out.append(b"".join(b"%.2X " % b for b in f["bytes"][: f["nbytes"]]))
        if f["complete"]:
            out.append(b"\n")
    return b"".join(out)


# ------------------------------------------------------------------ synthetic captures
class SynthParams(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("sample_rate", C.c_uint32), ("carrier_step", C.c_uint32), ("phase0", C.c_uint32),
        ("mod_index", C.c_uint32), ("amplitude", C.c_int32), ("noise_gain", C.c_int32), ("seed", C.c_uint64),
        ("signal_start", C.c_uint64), ("signal_end", C.c_uint64), ("doppler_q32", C.c_int64), ("env_floor_q15", C.c_uint32),
        ("fade_q15", C.c_uint32), ("fade_start", C.c_uint64), ("fade_len", C.c_uint64),
    ]


_synth = None


def synth_lib():
    global _synth
    if _synth is None:
        if not os.path.exists(LIBSYNTH_PATH):
            raise PdtError(f"{LIBSYNTH_PATH} is missing: run `make`")
        S = C.CDLL(LIBSYNTH_PATH)
        S.pdt_synth_default_params.argtypes = [C.POINTER(SynthParams), C.c_int, C.c_uint32, C.c_double, C.c_uint64]
        S.pdt_synth_default_params.restype = None
        S.pdt_synth_set_pass.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_double]
        S.pdt_synth_set_pass.restype = None
        S.pdt_synth_fill.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.c_uint64, C.c_void_p]
        S.pdt_synth_fill.restype = None
        S.pdt_synth_poes_frame.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.c_void_p]
        S.pdt_synth_poes_frame.restype = None
        S.pdt_synth_argos_payload.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.c_void_p]
        S.pdt_synth_argos_payload.restype = None
        S.pdt_synth_sine_table.restype = C.POINTER(C.c_int16)
        S.pdt_synth_wav_header.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        S.pdt_synth_wav_header.restype = None
        _synth = S
    return _synth


def synth_params(kind: int, sample_rate: int, f0_hz: float = 1000.0, seed: int = 1234) -> SynthParams:
    p = SynthParams()
    synth_lib().pdt_synth_default_params(C.byref(p), kind, sample_rate, f0_hz, seed)
    return p


def synth_capture(kind: int, sample_rate: int, seconds: float, f0_hz: float | None = None, seed: int = 1234,
                  start: int = 0) -> np.ndarray:
    """int16[n,2] synthetic POES (kind 0) / ARGOS (kind 1) capture; bit-reproducible everywhere."""
    if f0_hz is None:
        f0_hz = 1000.0 if kind == 0 else 120.0
    p = synth_params(kind, sample_rate, f0_hz, seed)
    n = int(round(seconds * sample_rate))
    out = np.zeros((n, 2), dtype="<i2")
    synth_lib().pdt_synth_fill(C.byref(p), start, n, out.ctypes.data)
    return out


def synth_poes_frame(p: SynthParams, fr: int) -> np.ndarray:
    out = np.zeros(104, dtype=np.uint8)
    synth_lib().pdt_synth_poes_frame(C.byref(p), fr, out.ctypes.data)
    return out


def synth_argos_payload(p: SynthParams, burst: int) -> np.ndarray:
    out = np.zeros(7, dtype=np.uint8)
    synth_lib().pdt_synth_argos_payload(C.byref(p), burst, out.ctypes.data)
    return out


def write_wav(path: str, sample_rate: int, iq: np.ndarray):
    hdr = C.create_string_buffer(44)
    synth_lib().pdt_synth_wav_header(hdr, sample_rate, iq.shape[0])
    with open(path, "wb") as f:
        f.write(hdr.raw)
        f.write(np.ascontiguousarray(iq, dtype="<i2").tobytes())
