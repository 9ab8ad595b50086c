"""Multi-process (gloo, world_size 2 and 3, CPU) tests of the frame sharding + size all-gather.

The GPU kernels are replaced by the CPU oracle here (tests may use it); what is under test is the
host logic that makes N ranks produce byte-identical output to one rank.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, channels, track, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle
    from sela_amd import sharding
    from sela_amd.synth import synth_frames

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pcm = synth_frames(n_frames, channels, track)
        b, e = sharding.my_range(n_frames, rank, world)
        blob, offs, _ = oracle().encode_frames(pcm[b:e], threads=1)
        local_offsets = torch.from_numpy(offs.astype(np.int64))
        layout = sharding.gather_frame_sizes(local_offsets, n_frames, rank, world)
        lo, hi = layout.rank_byte_range(rank)
        assert hi - lo == len(blob)
        # every rank writes its bytes at its offset of one shared file (what the host does with pwrite)
        path = os.path.join(tmpdir, "frames.bin")
        if rank == 0:
            with open(path, "wb") as f:
                f.truncate(int(layout.frame_offsets[-1]))
        dist.barrier()
        with open(path, "r+b") as f:
            f.seek(lo)
            f.write(blob.tobytes())
        dist.barrier()
        np.save(os.path.join(tmpdir, f"offsets_{rank}.npy"), layout.frame_offsets)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,channels", [(2, 11, 2), (3, 7, 1), (2, 1, 2)])
def test_sharded_output_is_byte_identical_to_single_rank(tmp_path, world, n_frames, channels):
    from oracle_lib import oracle
    from sela_amd.synth import synth_frames

    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, channels, 5, str(tmp_path)), nprocs=world, join=True)
    pcm = synth_frames(n_frames, channels, 5)
    blob, offs, _ = oracle().encode_frames(pcm, threads=2)
    got = np.fromfile(tmp_path / "frames.bin", dtype=np.uint8)
    assert np.array_equal(got, blob)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"offsets_{r}.npy"), offs)


def test_partition_properties():
    from sela_amd import sharding

    for n in (0, 1, 7, 8, 9, 3875, 549365):
        for w in (1, 2, 3, 4, 8):
            parts = sharding.partition(n, w)
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


def test_single_rank_layout():
    from sela_amd import sharding

    lay = sharding.gather_frame_sizes(np.array([0, 10, 30, 34], np.uint64), 3, 0, 1)
    assert lay.frame_sizes.tolist() == [10, 20, 4] and lay.rank_byte_range(0) == (0, 34)


# ---- many tracks, one job (configs[3]) --------------------------------------------------------------------
TRACKS = [(5, 44100), (0, 48000), (3, 48000), (7, 96000), (1, 44100)]  # (frames, sample rate): unequal, one empty


def _multi_track_worker(rank, world, port, channels, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle
    from sela_amd import sharding
    from sela_amd.synth import synth_frames

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = [n for n, _ in TRACKS]
        job = np.concatenate([synth_frames(n, channels, 40 + i) for i, n in enumerate(frames)])
        b, e = sharding.my_range(len(job), rank, world)
        blob, offs, _ = oracle().encode_frames(job[b:e], threads=1)
        layout = sharding.gather_frame_sizes(torch.from_numpy(offs.astype(np.int64)), len(job), rank, world)
        # the rank that owns nothing of a track never touches its file; rank 0 creates all files with headers
        if rank == 0:
            starts = sharding.track_starts(frames)
            for i, (n, rate) in enumerate(TRACKS):
                size = sharding.SELA_HEADER_BYTES + int(layout.frame_offsets[starts[i + 1]] - layout.frame_offsets[starts[i]])
                with open(os.path.join(tmpdir, f"track{i}.sela"), "wb") as f:
                    f.write(sharding.sela_header(rate, 16, channels, n))
                    f.truncate(size)
        dist.barrier()
        local_base = int(layout.frame_offsets[b])
        for piece in sharding.rank_track_pieces(layout, frames, rank):
            lo = int(layout.frame_offsets[piece.job_frame]) - local_base
            with open(os.path.join(tmpdir, f"track{piece.track}.sela"), "r+b") as f:
                f.seek(piece.file_offset)
                f.write(blob[lo: lo + piece.n_bytes].tobytes())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_many_tracks_sharded_across_ranks(tmp_path, world):
    """Ranks own contiguous ranges of the flattened (track, frame) space, cut anywhere; every track's .sela
    file (header + frames, src/file/sela_file.cpp:105-137) must equal what one process writes."""
    from oracle_lib import oracle
    from sela_amd import sharding
    from sela_amd.synth import synth_frames

    channels = 2
    mp.spawn(_multi_track_worker, args=(world, _free_port(), channels, str(tmp_path)), nprocs=world, join=True)
    for i, (n, rate) in enumerate(TRACKS):
        blob, _, _ = oracle().encode_frames(synth_frames(n, channels, 40 + i), threads=1)
        want = sharding.sela_header(rate, 16, channels, n) + blob.tobytes()
        assert (tmp_path / f"track{i}.sela").read_bytes() == want, i
