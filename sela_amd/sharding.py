"""Frame sharding across the GPUs of a node.

Frames are independent units (the predictor restarts every frame, the stereo decision is intra-frame),
so a batch shards with NO data-path collective: rank r of W takes a contiguous, balanced range of the
flattened (track, frame) index space -- the reference's own static partition
(src/sela/encoder.cpp:58-73) with the remainder spread instead of dumped on the last worker.

The path has exactly one exchange step: every rank needs the compressed size of every frame to know
where its bytes land in the output file.  That is one small all-gather of u32-sized values (4 bytes x
frames; latency bound, nothing to do with xGMI link bandwidth), then an exclusive scan that every rank
computes locally.  `torch.distributed` is the transport: backend "nccl" is RCCL on ROCm (GPU tensors),
"gloo" is used by the CPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np


def partition(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous balanced ranges [begin, end) for every rank; the first n_items % world ranks get one more."""
    base, extra = divmod(n_items, world)
    out, begin = [], 0
    for r in range(world):
        end = begin + base + (1 if r < extra else 0)
        out.append((begin, end))
        begin = end
    return out


def my_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    return partition(n_items, world)[rank]


@dataclass
class FileLayout:
    """Where every frame of the whole job lands in the output frame stream."""
    frame_sizes: np.ndarray    # uint64 [n_frames] bytes of every frame, global order
    frame_offsets: np.ndarray  # uint64 [n_frames + 1] exclusive scan
    rank_ranges: List[Tuple[int, int]]

    def rank_byte_range(self, rank: int) -> Tuple[int, int]:
        b, e = self.rank_ranges[rank]
        return int(self.frame_offsets[b]), int(self.frame_offsets[e])


def gather_frame_sizes(local_offsets, n_frames_total: int, rank: int, world: int, group=None) -> FileLayout:
    """All-gather the per-frame compressed sizes.

    local_offsets: the [n_local + 1] frame offsets this rank's encoder produced (torch tensor on the
    device the process group communicates on -- cuda for nccl/RCCL, cpu for gloo -- or a numpy array
    when world == 1).  Returns the layout of the whole job, identical on every rank.
    """
    ranges = partition(n_frames_total, world)
    if world == 1:
        offs = np.asarray(local_offsets.cpu() if hasattr(local_offsets, "cpu") else local_offsets).astype(np.uint64)
        sizes = np.diff(offs)
        return FileLayout(sizes, offs - offs[0], ranges)
    import torch
    import torch.distributed as dist

    n_local = ranges[rank][1] - ranges[rank][0]
    assert local_offsets.numel() == n_local + 1
    max_local = max(e - b for b, e in ranges)
    sizes = (local_offsets[1:] - local_offsets[:-1]).to(torch.int64)
    padded = torch.zeros(max_local, dtype=torch.int64, device=local_offsets.device)
    padded[:n_local] = sizes
    gathered = torch.empty(world * max_local, dtype=torch.int64, device=local_offsets.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    g = gathered.cpu().numpy().reshape(world, max_local)
    all_sizes = np.concatenate([g[r, : ranges[r][1] - ranges[r][0]] for r in range(world)]).astype(np.uint64)
    offsets = np.zeros(n_frames_total + 1, np.uint64)
    np.cumsum(all_sizes, out=offsets[1:])
    return FileLayout(all_sizes, offsets, ranges)


def encode_sharded(pcm_local, n_frames_total: int, rank: int, world: int, encoder, group=None):
    """Encode this rank's frames (pcm_local: int16 cuda tensor [n_local, 2048, ch]) and exchange sizes.

    Returns (EncodedFrames of the local shard, FileLayout of the whole job).  The local bytes belong at
    layout.rank_byte_range(rank) of the job's frame stream (pwrite at that offset + 15-byte file header).
    """
    out = encoder.encode(pcm_local)
    layout = gather_frame_sizes(out.offsets, n_frames_total, rank, world, group)
    return out, layout


# ---- many tracks in one job (BASELINE.json configs[3]: 100 tracks over 8 GPUs) ----------------------------
# The job's frames are the tracks' frames back to back; ranks still own contiguous ranges of that one
# index space, so a rank's range may start or end in the middle of a track.  Each track becomes its own
# .sela file: 15-byte header (src/file/sela_file.cpp:108-114) + that track's frames.
SELA_HEADER_BYTES = 15


@dataclass
class TrackPiece:
    """The part of one track that one rank holds."""
    track: int
    first_frame: int       # within the track
    n_frames: int
    job_frame: int         # index of first_frame in the job's flattened frame space
    file_offset: int       # where the piece's bytes go in the track's .sela file (behind its header)
    n_bytes: int


def track_starts(track_frames: List[int]) -> np.ndarray:
    """Job-level index of every track's first frame (+ the total at the end)."""
    out = np.zeros(len(track_frames) + 1, np.int64)
    np.cumsum(np.asarray(track_frames, np.int64), out=out[1:])
    return out


def rank_track_pieces(layout: FileLayout, track_frames: List[int], rank: int) -> List[TrackPiece]:
    """Split the frame range of `rank` at track boundaries and place every piece in its track's file."""
    starts = track_starts(track_frames)
    assert int(starts[-1]) == len(layout.frame_sizes)
    begin, end = layout.rank_ranges[rank]
    pieces = []
    t = int(np.searchsorted(starts, begin, side="right")) - 1
    while begin < end:
        while int(starts[t + 1]) <= begin:  # skip empty tracks
            t += 1
        stop = min(end, int(starts[t + 1]))
        pieces.append(TrackPiece(
            track=t, first_frame=begin - int(starts[t]), n_frames=stop - begin, job_frame=begin,
            file_offset=SELA_HEADER_BYTES + int(layout.frame_offsets[begin] - layout.frame_offsets[int(starts[t])]),
            n_bytes=int(layout.frame_offsets[stop] - layout.frame_offsets[begin])))
        begin = stop
    return pieces


def sela_header(sample_rate: int, bits_per_sample: int, channels: int, n_frames: int) -> bytes:
    import struct

    return b"SeLa" + struct.pack("<IHBI", sample_rate, bits_per_sample, channels, n_frames)
