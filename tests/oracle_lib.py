"""ctypes bindings for the parity checkers (TEST INFRASTRUCTURE ONLY).

`oracle()` -> oracle/libsela_oracle.so (our CPU restatement, built on demand with gcc).
`reference()` -> oracle/_ref/libsela_ref.so (the unmodified reference; prebuilt in the build
container by `make -C oracle ref`; returns None where it does not exist).

Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


class LpcTrace(C.Structure):
    _fields_ = [("mean", C.c_double), ("ac", C.c_double * 101), ("k", C.c_double * 100)]


def build_oracle() -> str:
    override = os.environ.get("SELA_ORACLE_LIB")  # tests/test_sanitizers.py: the sanitizer build of the same source
    if override:
        return override
    path = os.path.join(ORACLE_DIR, "libsela_oracle.so")
    src = os.path.join(ORACLE_DIR, "sela_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libsela_oracle.so"], stdout=subprocess.DEVNULL)
    return path


class Oracle:
    """Uniform wrapper over either library (prefix 'sela_oracle_' or 'ref_')."""

    def __init__(self, lib: C.CDLL, prefix: str):
        self.lib = lib
        self.prefix = prefix
        self.is_reference = prefix == "ref_"
        f = lambda name: getattr(lib, prefix + name)
        flags = [] if self.is_reference else [C.POINTER(C.c_uint32)]

        self._analyze = f("lpc_analyze")
        self._analyze.restype = C.c_int
        if self.is_reference:
            self._analyze.argtypes = [_i32p, C.c_int, _i32p, _i32p]
        else:
            self._analyze.argtypes = [_i32p, C.c_int, _i32p, _i64p, _i32p, C.POINTER(LpcTrace), C.POINTER(C.c_uint32)]
        self._coeffs = f("lpc_coeffs")
        self._coeffs.restype = None
        self._coeffs.argtypes = [C.c_int, _i32p, _i64p] + flags
        self._synth = f("lpc_synth")
        self._synth.restype = None
        self._synth.argtypes = [C.c_int, _i32p, _i32p, C.c_int, _i32p] + flags
        self._renc = f("rice_encode")
        self._renc.restype = C.c_int
        self._renc.argtypes = [_i32p, C.c_int, C.POINTER(C.c_uint32), _u32p, C.c_int] + flags
        self._rdec = f("rice_decode")
        self._rdec.restype = None
        self._rdec.argtypes = [_u32p, C.c_int, C.c_int, C.c_uint32, _i32p] + flags
        self._fenc = f("frame_encode")
        self._fenc.restype = C.c_size_t
        self._fenc.argtypes = [_i16p, C.c_uint32, C.c_uint32, _u8p] + flags
        self._fdec = f("frame_decode")
        self._fdec.restype = C.c_size_t
        self._fdec.argtypes = [_u8p, C.c_uint32, _i16p] + flags
        self._fenc32 = f("frame_encode_i32")
        self._fenc32.restype = C.c_size_t
        self._fenc32.argtypes = [_i32p, C.c_uint32, C.c_uint32, _u8p] + flags
        self._fencr = f("frame_encode_ragged")
        self._fencr.restype = C.c_size_t
        self._fencr.argtypes = [_i32p, _u32p, C.c_uint32, _u8p] + flags
        self._fdec32 = f("frame_decode_i32")
        self._fdec32.restype = C.c_size_t
        self._fdec32.argtypes = [_u8p, C.c_uint32, _i32p, C.c_uint32, _u32p] + flags
        self._encmt = f("encode_frames_mt")
        self._encmt.restype = C.c_double
        self._encmt.argtypes = [_i16p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _u8p, _u64p]
        self._decmt = f("decode_frames_mt")
        self._decmt.restype = C.c_double
        self._decmt.argtypes = [_u8p, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _i16p]

    def _fl(self):
        return [] if self.is_reference else [C.byref(C.c_uint32(0))]

    # -- per-stage -------------------------------------------------------------------------
    def lpc_analyze(self, samples, with_trace=False):
        s = np.ascontiguousarray(samples, dtype=np.int32)
        q = np.zeros(100, np.int32)
        r = np.zeros(len(s), np.int32)
        if self.is_reference:
            order = self._analyze(s, len(s), q, r)
            return order, q[:order].copy(), r
        a = np.zeros(101, np.int64)
        tr = LpcTrace()
        fl = C.c_uint32(0)
        order = self._analyze(s, len(s), q, a, r, C.byref(tr), C.byref(fl))
        if with_trace:
            return order, q[:order].copy(), r, a[: order + 1].copy(), tr, fl.value
        return order, q[:order].copy(), r

    def lpc_coeffs(self, order, q):
        qq = np.zeros(100, np.int32)
        qq[: len(q)] = q
        a = np.zeros(101, np.int64)
        self._coeffs(order, qq, a, *self._fl())
        return a[: order + 1].copy()

    def lpc_synth(self, order, q, residues):
        qq = np.zeros(100, np.int32)
        qq[: len(q)] = q
        r = np.ascontiguousarray(residues, dtype=np.int32)
        s = np.zeros(len(r), np.int32)
        self._synth(order, qq, r, len(r), s, *self._fl())
        return s

    def rice_encode(self, values):
        v = np.ascontiguousarray(values, dtype=np.int32)
        cap = 64 + 2 * len(v) * 4
        while True:
            words = np.zeros(cap, np.uint32)
            k = C.c_uint32(0)
            nw = self._renc(v, len(v), C.byref(k), words, cap, *self._fl())
            if nw >= 0 or cap > (1 << 24):
                break
            cap *= 8  # very long unary runs
        assert nw >= 0
        return k.value, words[:nw].copy()

    def rice_decode(self, words, n, k):
        w = np.ascontiguousarray(words, dtype=np.uint32)
        out = np.zeros(n, np.int32)
        self._rdec(w, len(w), n, k, out, *self._fl())
        return out

    # -- per-frame --------------------------------------------------------------------------
    def frame_encode(self, pcm):
        """pcm: int16 [n, channels] -> bytes of the on-disk frame."""
        p = np.ascontiguousarray(pcm, dtype=np.int16)
        n, ch = p.shape
        out = np.zeros(4 + ch * (12 + 4 * 128 + 8 * n + 256), np.uint8)
        used = self._fenc(p, ch, n, out, *self._fl())
        return out[:used].tobytes()

    def frame_decode(self, blob, channels, n=2048):
        b = np.frombuffer(blob, dtype=np.uint8).copy()
        pcm = np.zeros((n, channels), np.int16)
        used = self._fdec(b, channels, pcm, *self._fl())
        return pcm, used

    def frame_encode_i32(self, planar):
        """planar: int32 [channels, n] (data::WavFrame.samples) -> bytes of the on-disk frame."""
        p = np.ascontiguousarray(planar, dtype=np.int32)
        ch, n = p.shape
        out = np.zeros(4 + ch * (12 + 4 * 128 + 16 * n + 256), np.uint8)
        used = self._fenc32(p, ch, n, out, *self._fl())
        return out[:used].tobytes()

    def frame_encode_ragged(self, channels):
        """channels: list of int32 arrays of different lengths (data::WavFrame.samples) -> bytes of the on-disk frame."""
        chans = [np.ascontiguousarray(c, dtype=np.int32).ravel() for c in channels]
        flat = np.concatenate(chans) if chans else np.zeros(0, np.int32)
        lengths = np.array([len(c) for c in chans], np.uint32)
        out = np.zeros(4 + sum(12 + 4 * 128 + 16 * len(c) + 256 for c in chans), np.uint8)
        used = self._fencr(flat, lengths, len(chans), out, *self._fl())
        return out[:used].tobytes()

    def frame_decode_i32(self, blob, channels, stride=65535):
        """-> (list of int32 arrays, one per channel, as FrameDecoder::process returns them; bytes consumed)."""
        b = np.frombuffer(blob, dtype=np.uint8).copy()
        out = np.zeros((channels, stride), np.int32)
        counts = np.zeros(channels, np.uint32)
        used = self._fdec32(b, channels, out, stride, counts, *self._fl())
        return [out[c, : int(counts[c])].copy() for c in range(channels)], used

    # -- batch ------------------------------------------------------------------------------
    def encode_frames(self, pcm, threads=1):
        """pcm: int16 [n_frames, n, channels] -> (blob bytes, offsets uint64[n_frames+1], seconds)."""
        p = np.ascontiguousarray(pcm, dtype=np.int16)
        nf, n, ch = p.shape
        out = np.zeros(nf * (4 + ch * (12 + 4 * 128 + 8 * n + 256)) + 16, np.uint8)
        offs = np.zeros(nf + 1, np.uint64)
        secs = self._encmt(p, nf, ch, n, threads, out, offs)
        return out[: int(offs[nf])].copy(), offs, secs

    def decode_frames(self, blob, offsets, channels, n=2048, threads=1):
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        nf = len(offs) - 1
        pcm = np.zeros((nf, n, channels), np.int16)
        secs = self._decmt(b, offs, nf, channels, n, threads, pcm)
        return pcm, secs


_ORACLE = None
_REF = None


def oracle() -> Oracle:
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle(C.CDLL(build_oracle()), "sela_oracle_")
    return _ORACLE


def reference():
    """The real reference, or None when oracle/_ref/libsela_ref.so is not present."""
    global _REF
    if _REF is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libsela_ref.so")
        if not os.path.exists(path):
            return None
        try:
            _REF = Oracle(C.CDLL(path), "ref_")
        except OSError:
            return None
    return _REF
