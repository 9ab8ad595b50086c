"""GPU tests, by subject: more than one worker -- several processes / ranks / host threads sharing the one GPU of the test box (the multi-GPU
code paths of bench.py, sela_amd/sharding.py and sela::encodeFiles --devices), against the same digests as one worker."""
import hashlib
import os
import numpy as np
import pytest
from oracle_lib import oracle, reference
import subprocess
from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
import json
import sys
from sela_amd.synth import synth_frames, synth_pcm

from gpu_common import HOST, ROOT, _bench, _build, _sha_file, _write_wav, gpu, teams  # noqa: F401  (fixtures and helpers)

pytestmark = pytest.mark.gpu


def test_multi_gpu_dispatcher_with_two_workers_on_one_device(tmp_path):
    """sela::encodeBatch / decodeBatch with two host threads, both bound to device 0 (`--devices 0,0`): the
    flattened frame space is cut in two contiguous halves -- inside a track -- and the files must come out
    byte-identical to one worker's (src/sela/encoder.cpp:58-73 is the partition this replaces)."""
    _build()
    cli = os.path.join(HOST, "sela_mi355x")
    specs = [("a", 2, 44100, 9 * 2048 + 777), ("b", 2, 48000, 4 * 2048), ("c", 1, 96000, 2 * 2048 + 5), ("d", 2, 44100, 100),
             ("e", 2, 44100, 1100 * 2048), ("f", 1, 44100, 5 * 2048)]
    wavs = []
    for name, ch, rate, n in specs:
        p = tmp_path / f"{name}.wav"
        _write_wav(p, synth_pcm(n, ch, 80 + len(wavs)), rate)
        wavs.append(p)
    one, two, three, back1, back2 = (tmp_path / d for d in ("one", "two", "three", "back1", "back2"))
    for d in (one, two, three, back1, back2):
        d.mkdir()
    for out_dir, devs in ((one, "0"), (two, "0,0"), (three, "0,0,0")):
        r = subprocess.run([cli, "-E", str(out_dir), "--devices", devs] + [str(w) for w in wavs], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    for w in wavs:
        name = w.stem + ".sela"
        assert (two / name).read_bytes() == (one / name).read_bytes(), name
        assert (three / name).read_bytes() == (one / name).read_bytes(), name
    selas = [one / (w.stem + ".sela") for w in wavs]
    for out_dir, devs in ((back1, "0"), (back2, "0,0")):
        r = subprocess.run([cli, "-D", str(out_dir), "--devices", devs] + [str(s) for s in selas], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    for s in selas:
        name = s.stem + ".wav"
        assert (back2 / name).read_bytes() == (back1 / name).read_bytes(), name
    # a device that does not exist is an error, not a silent single-GPU run
    r = subprocess.run([cli, "-E", str(two), "--devices", "0,99", str(wavs[0])], capture_output=True, text=True)
    assert r.returncode == 1 and "device" in r.stderr


def test_sharded_encode_over_rccl_world_of_one(gpu, tmp_path):
    """sharding.encode_sharded with the `nccl` (= RCCL) backend on this GPU, world size 1: the N > 1 code path
    of bench.py and of a per-rank deployment, layout identical to the one-rank layout."""
    import torch
    import torch.distributed as dist

    from sela_amd import codec, sharding

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    if os.path.isdir("/sys/class/net/lo"): # (the bootstrap sockets over the loopback interface: the container's hostname may not resolve)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    import datetime

    dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device("cuda", 0))
    try:
        track_frames = [40, 0, 25]
        pcm = torch.cat([synth_frames_torch(n, 2, 90 + i, device="cuda") for i, n in enumerate(track_frames) if n])
        enc = codec.Encoder(pcm.shape[0], 2)
        out, layout = sharding.encode_sharded(pcm, pcm.shape[0], 0, 1, enc)
        # force the collective itself (encode_sharded short-cuts world == 1): one all-gather of the sizes
        sizes = (out.offsets[1:] - out.offsets[:-1]).to(torch.int64)
        gathered = torch.empty_like(sizes)
        dist.all_gather_into_tensor(gathered, sizes)
        torch.cuda.synchronize()
        frames, offsets = out.to_host()
        assert np.array_equal(gathered.cpu().numpy().astype(np.uint64), layout.frame_sizes)
        assert np.array_equal(layout.frame_offsets, offsets)
        ref_frames, ref_offsets, _ = oracle().encode_frames(pcm.cpu().numpy(), threads=8)
        assert np.array_equal(frames, ref_frames) and np.array_equal(offsets, ref_offsets)
        pieces = sharding.rank_track_pieces(layout, track_frames, 0)
        assert [(p.track, p.first_frame, p.n_frames) for p in pieces] == [(0, 0, 40), (2, 0, 25)]
        assert pieces[1].file_offset == sharding.SELA_HEADER_BYTES and pieces[1].n_bytes == int(offsets[65] - offsets[40])
    finally:
        dist.destroy_process_group()


def test_bench_line_of_the_multi_gpu_code_path():
    """bench.py as the driver runs it for N > 1 -- process group, the RCCL all-gather of the frame sizes inside the timed
    region, the layout checks -- on ONE GPU (SELA_BENCH_FORCE_EXCHANGE=1: a one-rank group): the headline (one track
    per GPU), the album block (BASELINE.json configs[3], layout against the reference's digest) and the decode10k block
    (configs[4]); every key the driver and the judge read is there."""
    line = _bench(["--steps", "2", "--warmup", "1", "--no-host-legs"], env={"SELA_BENCH_FORCE_EXCHANGE": "1"})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "lanes", "timed_outputs", "album", "decode10k"):
        assert key in line, key
    assert line["scaling"] == "weak" and line["n_gpus"] == 1 and line["value"] > 1000
    assert line["layout_matches_reference"] is True and line["digests_match_reference"] is True
    assert line["timed_outputs"]["equal_to_serial_step"] is True and line["timed_outputs"]["lanes_compared_with_serial_step"] == 2
    assert line["cpu_baseline"]["bit_exact_vs_gpu"] is True and line["cpu_baseline"]["cores"] >= 1
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    album = line["album"]
    assert album["layout_matches_reference"] is True and album["scaling"] == "strong" and album["config"]["frames_total"] == 549365
    assert album["timed_outputs"]["equal_to_serial_step"] is True
    d10 = line["decode10k"]
    assert d10["config"]["frames_total"] == 10000 and d10["bit_exact_vs_cpu_decode"] is True and d10["timed_outputs"]["equal_to_serial_step"] is True


def test_bench_album_as_the_headline():
    """`--workload album --steps 1 --warmup 0` with the exchange forced: the album as the line's own workload."""
    line = _bench(["--workload", "album", "--steps", "1", "--warmup", "0"], env={"SELA_BENCH_FORCE_EXCHANGE": "1"})
    assert line["layout_matches_reference"] is True and line["scaling"] == "strong" and line["config"]["frames_total"] == 549365
    assert "roofline" in line and line["value"] > 1000


def test_bench_relaunches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` without a launcher starts torch.distributed.run itself.  On a one-GPU box the second
    rank has no device to take: what matters here is that the command gets as far as the ranks (no assertion about
    WORLD_SIZE), and fails loudly rather than printing a line for fewer GPUs than asked."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra-legs", "--no-host-legs",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    import torch

    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["layout_matches_reference"] is True
    else:
        assert r.returncode != 0 and "AssertionError: --gpus" not in r.stderr
        assert not r.stdout.strip().startswith("{")


def test_bench_with_two_ranks_sharing_the_gpu():
    """The multi-rank logic of bench.py on a one-GPU box: two ranks under torch.distributed.run, both on GPU 0, talking
    over gloo (SELA_BENCH_RANKS_SHARE_GPU=1; RCCL refuses two ranks on one device).  Rank r's track (album track 3 r)
    against the reference's digests on every rank, the gathered layout of the two tracks, the album cut in two contiguous
    ranges (inside track 43) with its layout against the reference's, 10,000 frames decoded in two halves, one JSON line
    from rank 0 with n_gpus = 2.  Not a measurement: the two ranks share the device and the exchange goes through host
    memory."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SELA_BENCH_RANKS_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-host-legs", "--extra-steps", "1"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([t for t in r.stdout.strip().splitlines() if t.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frames_total"] == 2 * 3875
    assert line["layout_matches_reference"] is True and line["digests_match_reference"] is True
    assert line["timed_outputs"]["equal_to_serial_step"] is True and line["cpu_baseline"]["bit_exact_vs_gpu"] is True
    assert line["album"]["layout_matches_reference"] is True and line["album"]["config"]["frames_rank0"] in (274682, 274683)
    assert line["album"]["roundtrip_lossy_frames"] == 39  # (the reference's own lossy frames on the album, summed over the ranks)
    assert line["decode10k"]["config"]["frames_rank0"] == 5000 and line["decode10k"]["bit_exact_vs_cpu_decode"] is True


# ---- eight-way readiness on one GPU (no 8-GPU node has been available: SCALE_r01..r03 are `skipped` records) ----------------
def test_batch_verbs_eight_workers_on_album_tracks(gpu, tmp_path, album_digests):  # noqa: F811
    """Tracks 0..22 of BASELINE.json configs[3] (eight / eight / seven at 44.1 / 48 / 96 kHz, 123,803 frames; 24 tracks would
    put every eighth of the frame space exactly on a track boundary) through
    `sela_mi355x -E / -D --devices 0,0,0,0,0,0,0,0`: EIGHT workers bound to the one GPU, the frame space cut in eight
    contiguous ranges -- the reference's static partition (src/sela/encoder.cpp:58-73) with GPUs for threads -- seven cuts,
    most of them inside a track, every worker reading, coding and writing its own pieces.  Every .sela file and every decoded
    PCM against the unmodified reference's SHA-256s (tests/golden/album_digests.json)."""
    import shutil
    import subprocess
    import tempfile

    from sela_amd.synth import album_tracks, synth_frames_torch
    from gpu_common import _sha_file
    from test_host_cpp import HOST, _build, _write_wav

    _build()
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    work = tempfile.mkdtemp(dir=scratch)
    try:
        tracks = album_tracks()[:23]
        total = sum(frames for _, _, frames in tracks)
        cuts = [total * w // 8 for w in range(1, 8)]
        edges = np.cumsum([0] + [frames for _, _, frames in tracks])
        assert sum(1 for c in cuts if c not in edges) >= 5, "the cuts are supposed to fall inside tracks"
        wavs = []
        for track, rate, frames in tracks:
            pcm = synth_frames_torch(frames, 2, track, device="cuda").cpu().numpy().reshape(-1, 2)
            p = os.path.join(work, f"track{track:02d}.wav")
            _write_wav(p, pcm, rate)
            wavs.append(p)
        enc_dir, dec_dir = os.path.join(work, "enc"), os.path.join(work, "dec")
        os.mkdir(enc_dir), os.mkdir(dec_dir)
        cli = os.path.join(HOST, "sela_mi355x")
        devices = ",".join(["0"] * 8)
        r = subprocess.run([cli, "-E", enc_dir, "--devices", devices] + wavs, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for p in wavs:
            os.remove(p)
        selas = []
        for track, rate, frames in tracks:
            g = album_digests["tracks"][track]
            p = os.path.join(enc_dir, f"track{track:02d}.sela")
            assert os.path.getsize(p) == g["sela_bytes"], track
            assert _sha_file(p) == g["sela_sha256"], track
            selas.append(p)
        r = subprocess.run([cli, "-D", dec_dir, "--devices", devices] + selas, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for track, rate, frames in tracks:
            g = album_digests["tracks"][track]
            with open(os.path.join(dec_dir, f"track{track:02d}.wav"), "rb") as f:
                wav = f.read()
            assert len(wav) == 44 + frames * 2048 * 2 * 2
            assert hashlib.sha256(wav[44:]).hexdigest() == g["decoded_sha256"], track
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_bench_with_eight_ranks_sharing_the_gpu():
    """bench.py --gpus 8 as the driver launches it, on a one-GPU box: eight ranks under torch.distributed.run, all on GPU 0,
    talking over gloo (SELA_BENCH_RANKS_SHARE_GPU=1).  Rank r's track (album track 3 r) against the reference's digests on
    every rank; the gathered layout of the eight tracks; the album cut in eight contiguous ranges with the gathered layout
    against the reference's; 10,000 frames decoded in eighths; ONE JSON line, from rank 0, with n_gpus = 8.  Not a
    measurement (the ranks share the device): what it proves is that every code path of an 8-rank job runs and agrees."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SELA_BENCH_RANKS_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-host-legs", "--extra-steps", "1"],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [t for t in r.stdout.strip().splitlines() if t.startswith("{")]
    assert len(lines) == 1, "one JSON line, from rank 0"
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["frames_total"] == 8 * 3875
    assert line["layout_matches_reference"] is True and line["digests_match_reference"] is True
    assert line["timed_outputs"]["equal_to_serial_step"] is True and line["cpu_baseline"]["bit_exact_vs_gpu"] is True
    assert line["album"]["layout_matches_reference"] is True and line["album"]["config"]["frames_rank0"] in (68670, 68671)
    assert line["album"]["roundtrip_lossy_frames"] == 39  # (the reference's own lossy frames on the album, summed over the ranks)
    assert line["decode10k"]["config"]["frames_rank0"] == 1250 and line["decode10k"]["bit_exact_vs_cpu_decode"] is True
    assert line["decode10k"]["per_rank_share_8"] is None  # (the one-GPU line's prediction; an 8-rank line IS the thing)
