"""ctypes bindings of libsela_hip.so (the C ABI declared in include/sela_hip.h).

Importing this module does not need a GPU; calling into it does.  If the shared library has not
been built, `lib()` raises -- there is no fallback implementation.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsela_hip.so")

OK = 0
ERRORS = {-1: "ENODEV", -2: "EINVAL", -3: "ENOMEM", -4: "ECAPACITY", -5: "EFORMAT", -6: "ERANGE"}
SAMPLES_PER_FRAME = 2048

FLAG_Q_RANGE, FLAG_COEF_OVERFLOW, FLAG_RICE_RANGE, FLAG_RICE_OVERRUN, FLAG_WORDS_CAP, FLAG_BAD_FRAME, FLAG_INTERNAL, FLAG_SHORT_BLOCK = 1, 2, 4, 8, 16, 32, 64, 128

# every symbol include/sela_hip.h declares
EXPORTS = [
    "sela_hip_init", "sela_hip_shutdown", "sela_hip_thread_release", "sela_hip_last_error", "sela_hip_device_count",
    "sela_hip_signals_per_frame", "sela_hip_encode_workspace_bytes", "sela_hip_decode_workspace_bytes",
    "sela_hip_encode_bound_bytes", "sela_hip_encode_device", "sela_hip_decode_device",
    "sela_hip_encode", "sela_hip_decode", "sela_hip_index_frames",
    "sela_hip_enable_kernel_timing", "sela_hip_kernel_times",
    "sela_hip_lpc_encode", "sela_hip_lpc_decode", "sela_hip_rice_encode", "sela_hip_rice_decode",
    "sela_hip_host_alloc", "sela_hip_host_free", "sela_hip_decode_max_channels",
    "sela_hip_encode_begin", "sela_hip_encode_feed", "sela_hip_encode_end",
    "sela_hip_decode_begin", "sela_hip_decode_feed", "sela_hip_decode_end",
    "sela_hip_encode_bound_bytes_n", "sela_hip_index_samples", "sela_hip_encode_i32", "sela_hip_decode_i32", "sela_hip_encode_ragged_i32",
    "sela_hip_lpc_encode_n", "sela_hip_lpc_decode_n",
]


# the test hooks include/sela_hip_debug.h declares (not part of the boundary)
DEBUG_EXPORTS = [
    "sela_hip_debug_phase_buffer", "sela_hip_debug_force_plain_fir", "sela_hip_debug_mean_workers", "sela_hip_debug_encode_teams", "sela_hip_debug_encode_kernel", "sela_hip_debug_encode_fused", "sela_hip_debug_priorities", "sela_hip_debug_priorities_adaptive", "sela_hip_debug_launches_alone", "sela_hip_debug_block_forms", "sela_hip_debug_encode_hashes", "sela_hip_debug_standard_first", "sela_hip_debug_standard_chunks", "sela_hip_debug_segment_subframes", "sela_hip_debug_generic_wrap_taps", "sela_hip_debug_encode_split", "sela_hip_debug_launches_split", "sela_hip_debug_keep_both_candidates", "sela_hip_debug_stage_wait",
    "sela_hip_debug_reissued_feeds", "sela_hip_debug_contexts_created", "sela_hip_debug_decode_recurrence",
]


class Trace(C.Structure):
    """sela_hip_trace"""
    _fields_ = [
        ("mean", C.c_double), ("ac", C.c_double * 101), ("k", C.c_double * 100), ("a", C.c_int64 * 101),
        ("q", C.c_int32 * 100), ("order", C.c_int32), ("coef_k", C.c_uint32), ("coef_words", C.c_uint32),
        ("res_k", C.c_uint32), ("res_words", C.c_uint32), ("flags", C.c_uint32),
    ]


class SelaHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libsela_hip: {ERRORS.get(code, code)}: {message}")
        self.code = code


_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The SELA MI355X path has no CPU fallback.")
    # PyTorch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever is loaded first serves both
    # users; loaded second, torch would bring up a second HIP runtime in the process and this library's calls
    # would land in one that sees no device.  So: torch first, when there is a torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64p, sz = C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t
    L.sela_hip_init.argtypes = [C.c_int]
    L.sela_hip_init.restype = C.c_int
    L.sela_hip_shutdown.argtypes = []
    L.sela_hip_shutdown.restype = None
    L.sela_hip_thread_release.argtypes = []
    L.sela_hip_thread_release.restype = None
    L.sela_hip_last_error.argtypes = []
    L.sela_hip_last_error.restype = C.c_char_p
    L.sela_hip_device_count.argtypes = []
    L.sela_hip_device_count.restype = C.c_int
    L.sela_hip_signals_per_frame.argtypes = [u32]
    L.sela_hip_signals_per_frame.restype = u32
    for name in ("sela_hip_encode_workspace_bytes", "sela_hip_decode_workspace_bytes", "sela_hip_encode_bound_bytes"):
        getattr(L, name).argtypes = [u32, u32]
        getattr(L, name).restype = sz
    L.sela_hip_encode_device.argtypes = [vp, u32, u32, vp, sz, u64p, vp, vp, sz, vp, vp]
    L.sela_hip_encode_device.restype = C.c_int
    L.sela_hip_decode_device.argtypes = [vp, u64p, u32, u32, vp, vp, vp, sz, vp]
    L.sela_hip_decode_device.restype = C.c_int
    L.sela_hip_encode.argtypes = [vp, u32, u32, u32, vp, sz, vp]
    L.sela_hip_encode.restype = C.c_int
    L.sela_hip_decode.argtypes = [vp, vp, u32, u32, vp]
    L.sela_hip_decode.restype = C.c_int
    L.sela_hip_index_frames.argtypes = [vp, sz, u32, u32, vp]
    L.sela_hip_index_frames.restype = u32
    L.sela_hip_enable_kernel_timing.argtypes = [C.c_int]
    L.sela_hip_enable_kernel_timing.restype = None
    L.sela_hip_kernel_times.argtypes = [C.POINTER(C.c_float), C.c_int]
    L.sela_hip_kernel_times.restype = C.c_int
    L.sela_hip_lpc_encode.argtypes = [vp, u32, vp, vp, vp]
    L.sela_hip_lpc_decode.argtypes = [vp, vp, vp, u32, vp, vp]
    L.sela_hip_rice_encode.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    L.sela_hip_rice_decode.argtypes = [vp, vp, vp, vp, u32, vp]
    for name in ("sela_hip_lpc_encode", "sela_hip_lpc_decode", "sela_hip_rice_encode", "sela_hip_rice_decode"):
        getattr(L, name).restype = C.c_int
    L.sela_hip_encode_bound_bytes_n.argtypes = [u32, u32, u32]
    L.sela_hip_encode_bound_bytes_n.restype = sz
    L.sela_hip_index_samples.argtypes = [vp, vp, u32, u32, vp]
    L.sela_hip_index_samples.restype = u32
    L.sela_hip_encode_i32.argtypes = [vp, u32, u32, u32, vp, sz, vp]
    L.sela_hip_decode_i32.argtypes = [vp, vp, u32, u32, vp, u32, vp]
    L.sela_hip_encode_ragged_i32.argtypes = [vp, vp, u32, vp, sz, vp]
    L.sela_hip_lpc_encode_n.argtypes = [vp, u32, u32, vp, vp, vp]
    L.sela_hip_lpc_decode_n.argtypes = [vp, vp, vp, u32, u32, vp, vp]
    for name in ("sela_hip_encode_i32", "sela_hip_decode_i32", "sela_hip_encode_ragged_i32", "sela_hip_lpc_encode_n", "sela_hip_lpc_decode_n"):
        getattr(L, name).restype = C.c_int
    L.sela_hip_debug_phase_buffer.argtypes = [C.c_void_p]
    L.sela_hip_debug_phase_buffer.restype = None
    L.sela_hip_debug_force_plain_fir.argtypes = [C.c_int]
    L.sela_hip_debug_force_plain_fir.restype = None
    L.sela_hip_debug_mean_workers.argtypes = [C.c_int]
    L.sela_hip_debug_mean_workers.restype = None
    L.sela_hip_debug_encode_teams.argtypes = [C.c_int]
    L.sela_hip_debug_encode_teams.restype = None
    L.sela_hip_debug_encode_fused.argtypes = [C.c_int]
    L.sela_hip_debug_encode_fused.restype = None
    L.sela_hip_debug_priorities.argtypes = [C.c_uint32]
    L.sela_hip_debug_priorities.restype = None
    L.sela_hip_debug_priorities_adaptive.argtypes = []
    L.sela_hip_debug_priorities_adaptive.restype = None
    L.sela_hip_debug_launches_alone.argtypes = []
    L.sela_hip_debug_launches_alone.restype = C.c_int
    L.sela_hip_debug_encode_hashes.argtypes = [C.c_int]
    L.sela_hip_debug_encode_hashes.restype = None
    L.sela_hip_debug_standard_first.argtypes = [C.c_int]
    L.sela_hip_debug_standard_first.restype = None
    L.sela_hip_debug_standard_chunks.argtypes = []
    L.sela_hip_debug_standard_chunks.restype = C.c_int
    L.sela_hip_debug_segment_subframes.argtypes = []
    L.sela_hip_debug_segment_subframes.restype = C.c_longlong
    L.sela_hip_debug_generic_wrap_taps.argtypes = [C.c_int]
    L.sela_hip_debug_generic_wrap_taps.restype = None
    L.sela_hip_debug_encode_split.argtypes = [C.c_int]
    L.sela_hip_debug_encode_split.restype = None
    L.sela_hip_debug_launches_split.argtypes = []
    L.sela_hip_debug_launches_split.restype = C.c_int
    L.sela_hip_debug_block_forms.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.sela_hip_debug_block_forms.restype = C.c_int
    L.sela_hip_debug_keep_both_candidates.argtypes = [C.c_int]
    L.sela_hip_debug_keep_both_candidates.restype = None
    L.sela_hip_debug_encode_kernel.argtypes = [C.c_uint32, C.c_uint32]
    L.sela_hip_debug_encode_kernel.restype = C.c_int
    L.sela_hip_debug_stage_wait.argtypes = [C.c_int]
    L.sela_hip_debug_stage_wait.restype = None
    L.sela_hip_debug_reissued_feeds.argtypes = []
    L.sela_hip_debug_reissued_feeds.restype = C.c_int
    L.sela_hip_debug_decode_recurrence.argtypes = [C.c_int]
    L.sela_hip_debug_decode_recurrence.restype = None
    L.sela_hip_debug_contexts_created.argtypes = []
    L.sela_hip_debug_contexts_created.restype = C.c_int
    L.sela_hip_host_alloc.argtypes = [sz]
    L.sela_hip_host_alloc.restype = C.c_void_p
    L.sela_hip_host_free.argtypes = [C.c_void_p]
    L.sela_hip_host_free.restype = None
    L.sela_hip_decode_max_channels.argtypes = []
    L.sela_hip_decode_max_channels.restype = u32
    L.sela_hip_encode_begin.argtypes = [C.POINTER(C.c_void_p), u32, u32, vp, sz, vp]
    L.sela_hip_encode_feed.argtypes = [vp, vp, u32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.sela_hip_encode_end.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.sela_hip_decode_begin.argtypes = [C.POINTER(C.c_void_p), u32, u32, vp]
    L.sela_hip_decode_feed.argtypes = [vp, vp, vp, u32, C.POINTER(C.c_uint32)]
    L.sela_hip_decode_end.argtypes = [vp, C.POINTER(C.c_uint32)]
    for name in ("sela_hip_encode_begin", "sela_hip_encode_feed", "sela_hip_encode_end", "sela_hip_decode_begin",
                 "sela_hip_decode_feed", "sela_hip_decode_end"):
        getattr(L, name).restype = C.c_int
    _LIB = L
    return L


def kernel_times(capacity: int = 4):
    """Durations (ms) of the kernels of the calling thread's last *_device call (timing must be enabled)."""
    buf = (C.c_float * capacity)()
    n = lib().sela_hip_kernel_times(buf, capacity)
    return [float(buf[i]) for i in range(n)]


def check(rc: int) -> None:
    if rc != OK:
        raise SelaHipError(rc, lib().sela_hip_last_error().decode("utf-8", "replace"))
