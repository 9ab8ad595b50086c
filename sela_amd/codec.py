"""Torch/numpy wrappers over the C ABI.  PyTorch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi

BLOCK = capi.SAMPLES_PER_FRAME


@dataclass
class EncodedFrames:
    """Device-resident result of an encode: the .sela frame byte stream + per-frame offsets."""
    frames: "torch.Tensor"       # uint8 [capacity]; the first offsets[-1] bytes are valid
    offsets: "torch.Tensor"      # int64 [n_frames + 1] (bit pattern of the library's uint64)
    status: "torch.Tensor"       # int32 [4]
    n_frames: int
    channels: int

    def total_bytes(self) -> int:
        return int(self.offsets[-1].item())

    def check(self) -> None:
        st = self.status.cpu().numpy().view(np.uint32)
        bad = int(st[0]) & (capi.FLAG_WORDS_CAP | capi.FLAG_RICE_RANGE | capi.FLAG_COEF_OVERFLOW)
        if bad:
            raise capi.SelaHipError(-6, f"a block left the range the .sela format can carry (flags 0x{int(st[0]):x})")
        if int(st[1]):
            raise capi.SelaHipError(-4, "frame buffer too small")

    def to_host(self):
        self.check()
        n = self.total_bytes()
        return self.frames[:n].cpu().numpy(), self.offsets.cpu().numpy().view(np.uint64)


class Encoder:
    """Reusable device buffers for encoding batches of up to `max_frames` frames on the current device."""

    def __init__(self, max_frames: int, channels: int, device=None, with_trace: bool = False):
        import torch

        self.torch = torch
        self.lib = capi.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_frames, self.channels = max_frames, channels
        self.n_sig = int(self.lib.sela_hip_signals_per_frame(channels))
        ws = int(self.lib.sela_hip_encode_workspace_bytes(max_frames, channels))
        # the algorithmic output never exceeds the input for audio; keep the certain bound small by
        # default (2x PCM) and let callers who want certainty pass frames_capacity explicitly
        self.capacity = max(2 * max_frames * BLOCK * channels * 2, 4096)
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(ws, dtype=torch.uint8, device=self.device)
            self.frames = torch.empty(self.capacity, dtype=torch.uint8, device=self.device)
            self.offsets = torch.empty(max_frames + 1, dtype=torch.int64, device=self.device)
            self.status = torch.zeros(4, dtype=torch.int32, device=self.device)
            self.trace = (torch.zeros(max_frames * self.n_sig * C.sizeof(capi.Trace), dtype=torch.uint8, device=self.device)
                          if with_trace else None)

    def encode(self, pcm, status=None) -> EncodedFrames:
        """pcm: int16 cuda tensor [n_frames, 2048, channels] (contiguous).  Asynchronous on the current stream.
        status: where this call leaves its four status words (int32 cuda tensor [4]); default: the encoder's own."""
        torch = self.torch
        status = self.status if status is None else status
        assert status.dtype == torch.int32 and status.numel() == 4 and status.is_cuda and status.is_contiguous()
        assert pcm.dtype == torch.int16 and pcm.is_cuda and pcm.is_contiguous()
        n_frames = pcm.shape[0]
        assert pcm.shape[1] == BLOCK and pcm.shape[2] == self.channels and n_frames <= self.max_frames
        stream = torch.cuda.current_stream(self.device).cuda_stream
        capi.check(self.lib.sela_hip_encode_device(
            pcm.data_ptr(), n_frames, self.channels, self.frames.data_ptr(), self.capacity, self.offsets.data_ptr(),
            status.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(),
            self.trace.data_ptr() if self.trace is not None else None, stream))
        return EncodedFrames(self.frames, self.offsets[: n_frames + 1], status, n_frames, self.channels)

    def traces(self, n_frames: int):
        raw = self.trace[: n_frames * self.n_sig * C.sizeof(capi.Trace)].cpu().numpy().tobytes()
        return (capi.Trace * (n_frames * self.n_sig)).from_buffer_copy(raw)


class Decoder:
    def __init__(self, max_frames: int, channels: int, device=None):
        import torch

        self.torch = torch
        self.lib = capi.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_frames, self.channels = max_frames, channels
        with torch.cuda.device(self.device):
            self.pcm = torch.empty((max_frames, BLOCK, channels), dtype=torch.int16, device=self.device)
            self.status = torch.zeros(4, dtype=torch.int32, device=self.device)
            ws = int(self.lib.sela_hip_decode_workspace_bytes(max_frames, channels))
            self.workspace = torch.empty(ws, dtype=torch.uint8, device=self.device)

    def decode(self, frames, offsets, n_frames: int, status=None):
        """frames: uint8 cuda tensor, offsets: int64 cuda tensor [n_frames+1].  Asynchronous.
        status: where this call leaves its status words (int32 cuda tensor [4]); default: the decoder's own."""
        torch = self.torch
        status = self.status if status is None else status
        assert status.dtype == torch.int32 and status.numel() == 4 and status.is_cuda and status.is_contiguous()
        assert frames.is_cuda and offsets.is_cuda and offsets.dtype == torch.int64 and n_frames <= self.max_frames
        stream = torch.cuda.current_stream(self.device).cuda_stream
        capi.check(self.lib.sela_hip_decode_device(
            frames.data_ptr(), offsets.data_ptr(), n_frames, self.channels, self.pcm.data_ptr(), status.data_ptr(),
            self.workspace.data_ptr(), self.workspace.numel(), stream))
        return self.pcm[:n_frames]

    def check(self) -> None:
        st = self.status.cpu().numpy().view(np.uint32)
        if int(st[0]) & capi.FLAG_BAD_FRAME:
            raise capi.SelaHipError(-5, f"malformed frame stream ({int(st[1])} bad frames)")
        if int(st[0]) & capi.FLAG_RICE_OVERRUN:
            raise capi.SelaHipError(-5, "a Rice stream ended before all its values were read")


# ---- host-pointer API on numpy arrays (what the C++ host calls) --------------------------------------
def encode_host(pcm: np.ndarray):
    """pcm: int16 [n_frames, n, channels] (n = 2048: the fast kernels; anything else in 1..65535: the any-length route)
    -> (frames uint8[...], offsets uint64[n_frames+1])."""
    lib = capi.lib()
    p = np.ascontiguousarray(pcm, dtype=np.int16)
    n_frames, n, ch = p.shape
    cap = max(2 * p.nbytes, 4096) if n == BLOCK else int(lib.sela_hip_encode_bound_bytes_n(n_frames, ch, n))
    frames = np.empty(cap, np.uint8)
    offs = np.zeros(n_frames + 1, np.uint64)
    capi.check(lib.sela_hip_encode(p.ctypes.data, n_frames, ch, n, frames.ctypes.data, cap, offs.ctypes.data))
    return frames[: int(offs[n_frames])].copy(), offs


def decode_host(frames: np.ndarray, offsets: np.ndarray, channels: int) -> np.ndarray:
    """-> int16 [n_frames, 2048, channels] for a stream of 2048-sample frames; for any other stream int16 [total samples,
    channels], frame f's samples from index_samples()[0][f] on."""
    lib = capi.lib()
    fr = np.ascontiguousarray(frames, dtype=np.uint8)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    n_frames = len(offs) - 1
    sample_offsets, largest = index_samples(fr, offs, channels)
    standard = largest == BLOCK and all(int(sample_offsets[f]) == f * BLOCK for f in range(n_frames + 1))
    total = n_frames * BLOCK if standard or largest == 0 else int(sample_offsets[n_frames])
    pcm = np.empty((max(total, n_frames * BLOCK, 1), channels), np.int16)  # (the fast kernels are tried first: sela_hip.h)
    capi.check(lib.sela_hip_decode(fr.ctypes.data, offs.ctypes.data, n_frames, channels, pcm.ctypes.data))
    return pcm[:total].reshape(n_frames, BLOCK, channels) if standard or largest == 0 else pcm[:total]


def index_samples(frames: np.ndarray, offsets: np.ndarray, channels: int):
    """-> (sample_offsets uint64[n_frames+1], largest samplesPerChannel of the stream)."""
    lib = capi.lib()
    fr = np.ascontiguousarray(frames, dtype=np.uint8)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    n_frames = len(offs) - 1
    so = np.zeros(n_frames + 1, np.uint64)
    largest = lib.sela_hip_index_samples(fr.ctypes.data, offs.ctypes.data, n_frames, channels, so.ctypes.data)
    return so, int(largest)


def encode_i32(samples: np.ndarray):
    """frame::FrameEncoder on data::WavFrame values: samples int32 [n_frames, channels, n] -> (frames uint8[...], offsets)."""
    lib = capi.lib()
    p = np.ascontiguousarray(samples, dtype=np.int32)
    n_frames, ch, n = p.shape
    cap = int(lib.sela_hip_encode_bound_bytes_n(n_frames, ch, n))
    frames = np.empty(max(cap, 16), np.uint8)
    offs = np.zeros(n_frames + 1, np.uint64)
    capi.check(lib.sela_hip_encode_i32(p.ctypes.data, n_frames, ch, n, frames.ctypes.data, cap, offs.ctypes.data))
    return frames[: int(offs[n_frames])].copy(), offs


def encode_ragged(channels) -> bytes:
    """frame::FrameEncoder on a data::WavFrame whose channels differ in length: list of int32 arrays -> the frame's bytes."""
    lib = capi.lib()
    chans = [np.ascontiguousarray(c, dtype=np.int32).ravel() for c in channels]
    flat = np.concatenate(chans)
    lengths = np.array([len(c) for c in chans], np.uint32)
    cap = 4 + sum(int(lib.sela_hip_encode_bound_bytes_n(1, 1, len(c))) for c in chans)
    out = np.empty(cap, np.uint8)
    used = C.c_size_t(0)
    capi.check(lib.sela_hip_encode_ragged_i32(flat.ctypes.data, lengths.ctypes.data, len(chans), out.ctypes.data, cap, C.byref(used)))
    return out[: used.value].tobytes()


def decode_i32(frames: np.ndarray, offsets: np.ndarray, channels: int, stride=None):
    """frame::FrameDecoder as it returns: -> list (per frame) of lists (per channel) of int32 arrays."""
    lib = capi.lib()
    fr = np.ascontiguousarray(frames, dtype=np.uint8)
    offs = np.ascontiguousarray(offsets, dtype=np.uint64)
    n_frames = len(offs) - 1
    if stride is None:
        stride = max(index_samples(fr, offs, channels)[1], 1)
    out = np.zeros((n_frames, channels, stride), np.int32)
    counts = np.zeros((n_frames, channels), np.uint32)
    capi.check(lib.sela_hip_decode_i32(fr.ctypes.data, offs.ctypes.data, n_frames, channels, out.ctypes.data, stride, counts.ctypes.data))
    return [[out[f, c, : int(counts[f, c])].copy() for c in range(channels)] for f in range(n_frames)]


def index_frames(frames: np.ndarray, n_frames: int, channels: int) -> np.ndarray:
    lib = capi.lib()
    fr = np.ascontiguousarray(frames, dtype=np.uint8)
    offs = np.zeros(n_frames + 1, np.uint64)
    found = lib.sela_hip_index_frames(fr.ctypes.data, fr.nbytes, n_frames, channels, offs.ctypes.data)
    return offs[: found + 1]


# ---- the stages on their own (sela_hip_lpc_* / sela_hip_rice_*: the reference's L1 classes, batched) -----------------------
def lpc_encode(samples: np.ndarray):
    """samples: int32 [n_blocks, len] -> (order int32[n], q int32[n, 100] (first order[i] entries valid), residues int32[n, len]).
    len = 2048 is sela_hip_lpc_encode (the frame kernels' analysis), any other length sela_hip_lpc_encode_n."""
    lib = capi.lib()
    s = np.ascontiguousarray(samples, dtype=np.int32)
    n = s.shape[0]
    assert s.ndim == 2
    length = s.shape[1]
    order, q, res = np.zeros(n, np.int32), np.full((n, 100), 0x5A5A5A5A, np.int32), np.zeros((n, length), np.int32)
    if length == BLOCK:
        capi.check(lib.sela_hip_lpc_encode(C.c_void_p(s.ctypes.data), n, C.c_void_p(order.ctypes.data), C.c_void_p(q.ctypes.data), C.c_void_p(res.ctypes.data)))
    else:
        capi.check(lib.sela_hip_lpc_encode_n(C.c_void_p(s.ctypes.data), n, length, C.c_void_p(order.ctypes.data), C.c_void_p(q.ctypes.data), C.c_void_p(res.ctypes.data)))
    return order, q, res


def lpc_encode_n(samples: np.ndarray):
    """Always the any-length kernels (also for 2048)."""
    lib = capi.lib()
    s = np.ascontiguousarray(samples, dtype=np.int32)
    n, length = s.shape
    order, q, res = np.zeros(n, np.int32), np.full((n, 100), 0x5A5A5A5A, np.int32), np.zeros((n, length), np.int32)
    capi.check(lib.sela_hip_lpc_encode_n(C.c_void_p(s.ctypes.data), n, length, C.c_void_p(order.ctypes.data), C.c_void_p(q.ctypes.data), C.c_void_p(res.ctypes.data)))
    return order, q, res


def lpc_decode_n(order: np.ndarray, q: np.ndarray, residues: np.ndarray, want_coefficients: bool = False):
    """sela_hip_lpc_decode_n: residues int32 [n_blocks, len] of any length."""
    lib = capi.lib()
    o = np.ascontiguousarray(order, dtype=np.int32)
    n = o.shape[0]
    qq = np.ascontiguousarray(q, dtype=np.int32).reshape(n, 100)
    r = np.ascontiguousarray(residues, dtype=np.int32).reshape(n, -1)
    out = np.zeros_like(r)
    coefs = np.zeros((n, 101), np.int64) if want_coefficients else None
    capi.check(lib.sela_hip_lpc_decode_n(C.c_void_p(o.ctypes.data), C.c_void_p(qq.ctypes.data), C.c_void_p(r.ctypes.data), n, r.shape[1],
                                         C.c_void_p(out.ctypes.data), C.c_void_p(coefs.ctypes.data) if coefs is not None else None))
    return (out, coefs) if want_coefficients else out


def lpc_decode(order: np.ndarray, q: np.ndarray, residues: np.ndarray, want_coefficients: bool = False):
    """The inverse: -> samples int32 [n_blocks, 2048] (and, if asked, the Q35 predictors int64 [n_blocks, 101])."""
    lib = capi.lib()
    o = np.ascontiguousarray(order, dtype=np.int32)
    n = o.shape[0]
    qq = np.ascontiguousarray(q, dtype=np.int32).reshape(n, 100)
    r = np.ascontiguousarray(residues, dtype=np.int32).reshape(n, BLOCK)
    out = np.zeros((n, BLOCK), np.int32)
    coefs = np.zeros((n, 101), np.int64) if want_coefficients else None
    capi.check(lib.sela_hip_lpc_decode(C.c_void_p(o.ctypes.data), C.c_void_p(qq.ctypes.data), C.c_void_p(r.ctypes.data), n, C.c_void_p(out.ctypes.data),
                                       C.c_void_p(coefs.ctypes.data) if coefs is not None else None))
    return (out, coefs) if want_coefficients else out


def rice_encode(streams):
    """streams: list of int32 arrays -> list of (k, words uint32[...]) as rice::RiceEncoder gives them."""
    lib = capi.lib()
    n = len(streams)
    vals = [np.ascontiguousarray(v, dtype=np.int32).ravel() for v in streams]
    voff = np.zeros(n + 1, np.uint64)
    voff[1:] = np.cumsum([len(v) for v in vals])
    flat = np.concatenate(vals) if n and voff[n] else np.zeros(0, np.int32)
    # room: a codeword is at most 1 + 19 + (|value| << 1 >> 19) bits... sized by a first call that only asks for the counts
    woff = np.zeros(n + 1, np.uint64)
    k, counts = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    rc = lib.sela_hip_rice_encode(C.c_void_p(flat.ctypes.data), C.c_void_p(voff.ctypes.data), n, C.c_void_p(k.ctypes.data), C.c_void_p(counts.ctypes.data),
                                  None, C.c_void_p(woff.ctypes.data))
    if rc not in (capi.OK, -4):
        capi.check(rc)
    woff[1:] = np.cumsum(counts.astype(np.uint64))
    words = np.zeros(int(woff[n]), np.uint32)
    capi.check(lib.sela_hip_rice_encode(C.c_void_p(flat.ctypes.data), C.c_void_p(voff.ctypes.data), n, C.c_void_p(k.ctypes.data), C.c_void_p(counts.ctypes.data),
                                        C.c_void_p(words.ctypes.data), C.c_void_p(woff.ctypes.data)))
    return [(int(k[i]), words[int(woff[i]): int(woff[i + 1])].copy()) for i in range(n)]


def rice_decode(streams):
    """streams: list of (k, words uint32 array, count) -> list of int32 arrays (rice::RiceDecoder)."""
    lib = capi.lib()
    n = len(streams)
    ws = [np.ascontiguousarray(w, dtype=np.uint32).ravel() for _, w, _ in streams]
    woff, voff = np.zeros(n + 1, np.uint64), np.zeros(n + 1, np.uint64)
    woff[1:] = np.cumsum([len(w) for w in ws])
    voff[1:] = np.cumsum([c for _, _, c in streams])
    flat = np.concatenate(ws) if n and woff[n] else np.zeros(0, np.uint32)
    k = np.array([kk for kk, _, _ in streams], np.uint32)
    out = np.zeros(int(voff[n]), np.int32)
    capi.check(lib.sela_hip_rice_decode(C.c_void_p(flat.ctypes.data), C.c_void_p(woff.ctypes.data), C.c_void_p(k.ctypes.data), C.c_void_p(voff.ctypes.data), n,
                                        C.c_void_p(out.ctypes.data)))
    return [out[int(voff[i]): int(voff[i + 1])].copy() for i in range(n)]
