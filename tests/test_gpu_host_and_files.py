"""GPU tests, by subject: the C++ host (host/) -- the CLI verbs and the batch verbs against the file digests the unmodified reference wrote, the
reference's classes bound to the library and its unchanged main.cpp compiled against this host, files whose frames are not 2048 samples long."""
import hashlib
import os
import pytest
from oracle_lib import oracle, reference
import subprocess
from oracle_lib import oracle
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm
import generic_cases as gc
from sela_amd.synth import synth_frames, synth_pcm

from gpu_common import BOUND, HOST, MAIN_ON_HOST, _build, _sha_file, _write_wav, gpu, teams  # noqa: F401  (fixtures and helpers)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("label", ["config0_mono_10s", "config1_stereo_3min", "stereo_48k_tail", "three_channel_96k", "shorter_than_a_frame"])
def test_cli_files_match_reference_file_digests(tmp_path, file_digests, label):
    """`sela_mi355x -e` / `-d` (the streaming file-to-file path) write byte-identical files -- headers, dropped
    tail and all -- to the ones the reference's sela::Encoder + SelaFile::writeToFile and sela::Decoder +
    WavFile::writeToFile wrote (tests/golden/file_digests.json, made by oracle/_ref's ref_encode_file /
    ref_decode_file); so do the batch verbs -E / -D."""
    _build()
    d = file_digests[label]
    pcm = synth_pcm(d["samples_per_channel"], d["channels"], d["track"])
    wav, sela, back = tmp_path / "in.wav", tmp_path / "out.sela", tmp_path / "back.wav"
    _write_wav(wav, pcm, d["sample_rate"])
    assert _sha_file(wav) == d["wav_sha256"], "input drifted"
    cli = os.path.join(HOST, "sela_mi355x")
    r = subprocess.run([cli, "-e", str(wav), str(sela)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(sela) == d["sela_bytes"] and _sha_file(sela) == d["sela_sha256"]
    r = subprocess.run([cli, "-d", str(sela), str(back)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(back) == d["decoded_wav_bytes"] and _sha_file(back) == d["decoded_wav_sha256"]
    batch = tmp_path / "batch"
    batch.mkdir()
    r = subprocess.run([cli, "-E", str(batch), str(wav)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _sha_file(batch / "in.sela") == d["sela_sha256"]
    r = subprocess.run([cli, "-D", str(batch), str(batch / "in.sela")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _sha_file(batch / "in.wav") == d["decoded_wav_sha256"]


def test_cli_verbs_give_the_same_files_every_time(tmp_path, file_digests):
    """The host pipeline is copies, kernels and host hand-overs on four streams: a hand-over that is only almost right
    shows once in a while, not every time.  Six passes of each file verb over the 3-minute track, in fresh processes,
    every output against the reference's SHA-256."""
    _build()
    d = file_digests["config1_stereo_3min"]
    wav = tmp_path / "in.wav"
    _write_wav(wav, synth_pcm(d["samples_per_channel"], d["channels"], d["track"]), d["sample_rate"])
    cli = os.path.join(HOST, "sela_mi355x")
    for rep in range(6):
        sela, back, batch = tmp_path / f"out{rep}.sela", tmp_path / f"back{rep}.wav", tmp_path / f"batch{rep}"
        batch.mkdir()
        r = subprocess.run([cli, "-e", str(wav), str(sela)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert _sha_file(sela) == d["sela_sha256"], rep
        r = subprocess.run([cli, "-E", str(batch), str(wav)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert _sha_file(batch / "in.sela") == d["sela_sha256"], rep
        r = subprocess.run([cli, "-d", str(sela), str(back)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert _sha_file(back) == d["decoded_wav_sha256"], rep
        for f in (sela, back, batch / "in.sela"):
            os.remove(f)


def test_batch_verbs_on_album_tracks_match_reference_digests(gpu, tmp_path, album_digests):
    """Tracks 0..11 of BASELINE.json configs[3] (four each at 44.1 / 48 / 96 kHz, 66,120 frames) written as WAV files to
    tmpfs, `sela_mi355x -E --devices 0,0` (two workers on the one GPU: the frame space is cut inside track 7, every
    worker reads, codes and writes its own pieces) -> every .sela file's SHA-256 is the one the unmodified reference
    produced (tests/golden/album_digests.json); `-D --devices 0,0` back -> every decoded PCM's SHA-256 likewise."""
    _build()
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    import shutil
    import tempfile

    work = tempfile.mkdtemp(dir=scratch)
    try:
        tracks = album_tracks()[:12]
        wavs = []
        for track, rate, frames in tracks:
            pcm = synth_frames_torch(frames, 2, track, device="cuda").cpu().numpy().reshape(-1, 2)
            p = os.path.join(work, f"track{track:02d}.wav")
            _write_wav(p, pcm, rate)
            wavs.append(p)
        enc_dir, dec_dir = os.path.join(work, "enc"), os.path.join(work, "dec")
        os.mkdir(enc_dir), os.mkdir(dec_dir)
        cli = os.path.join(HOST, "sela_mi355x")
        r = subprocess.run([cli, "-E", enc_dir, "--devices", "0,0"] + wavs, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        selas = []
        for track, rate, frames in tracks:
            g = album_digests["tracks"][track]
            p = os.path.join(enc_dir, f"track{track:02d}.sela")
            assert os.path.getsize(p) == g["sela_bytes"], track
            assert _sha_file(p) == g["sela_sha256"], track
            selas.append(p)
        r = subprocess.run([cli, "-D", dec_dir, "--devices", "0,0"] + selas, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for track, rate, frames in tracks:
            g = album_digests["tracks"][track]
            with open(os.path.join(dec_dir, f"track{track:02d}.wav"), "rb") as f:
                wav = f.read()
            assert len(wav) == 44 + frames * 2048 * 2 * 2
            assert hashlib.sha256(wav[44:]).hexdigest() == g["decoded_sha256"], track
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_batch_verbs_small_tracks_in_one_job_and_three_workers(tmp_path):
    """Many small tracks (they are read into one buffer and coded as one job), a track shorter than a frame, two channel
    counts, three workers on one device: every file equals what `-e` / `-d` write for it alone."""
    _build()
    cli = os.path.join(HOST, "sela_mi355x")
    specs = [("a", 2, 44100, 9 * 2048 + 777), ("b", 2, 48000, 4 * 2048), ("c", 1, 96000, 2 * 2048 + 5), ("d", 2, 44100, 100),
             ("e", 2, 44100, 300 * 2048), ("f", 1, 44100, 5 * 2048), ("g", 2, 44100, 2048), ("h", 2, 44100, 37 * 2048 + 1), ("i", 2, 8000, 1500 * 2048)]
    wavs = []
    for k, (name, ch, rate, n) in enumerate(specs):
        p = tmp_path / f"{name}.wav"
        _write_wav(p, synth_pcm(n, ch, 120 + k), rate)
        wavs.append(p)
    single, batch, back1, back3 = (tmp_path / d for d in ("single", "batch", "back1", "back3"))
    for d in (single, batch, back1, back3):
        d.mkdir()
    for w in wavs:
        r = subprocess.run([cli, "-e", str(w), str(single / (w.stem + ".sela"))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([cli, "-E", str(batch), "--devices", "0,0,0", "--io-threads", "5"] + [str(w) for w in wavs], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for w in wavs:
        name = w.stem + ".sela"
        assert (batch / name).read_bytes() == (single / name).read_bytes(), name
    selas = [batch / (w.stem + ".sela") for w in wavs]
    for s in selas:
        r = subprocess.run([cli, "-d", str(s), str(back1 / (s.stem + ".wav"))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([cli, "-D", str(back3), "--devices", "0,0,0"] + [str(s) for s in selas], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for s in selas:
        name = s.stem + ".wav"
        assert (back3 / name).read_bytes() == (back1 / name).read_bytes(), name
    # a stream that stops early (a frame without its sync word): the batch decoder writes what the file decoder writes
    raw = bytearray((batch / "e.sela").read_bytes())
    offs = 15
    for _ in range(120):  # walk 120 frames to find the 121st's sync word
        assert raw[offs: offs + 4] == bytes([0x00, 0xFF, 0x55, 0xAA])
        p = offs + 4
        for _c in range(2):
            cw = raw[p + 4] | (raw[p + 5] << 8)
            p += 7 + 4 * cw
            rw = raw[p + 1] | (raw[p + 2] << 8)
            p += 5 + 4 * rw
        offs = p
    raw[offs] ^= 0xFF
    (tmp_path / "cut.sela").write_bytes(bytes(raw))
    r = subprocess.run([cli, "-d", str(tmp_path / "cut.sela"), str(tmp_path / "cut1.wav")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cutdir = tmp_path / "cutdir"
    cutdir.mkdir()
    r = subprocess.run([cli, "-D", str(cutdir), "--devices", "0,0", str(tmp_path / "cut.sela")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (cutdir / "cut.wav").read_bytes() == (tmp_path / "cut1.wav").read_bytes()
    assert len((tmp_path / "cut1.wav").read_bytes()) == 44 + 120 * 2048 * 2 * 2


@pytest.mark.parametrize("label", ["config0_mono_10s", "config1_stereo_3min", "stereo_48k_tail", "three_channel_96k", "shorter_than_a_frame"])
def test_reference_classes_bound_to_the_library_write_the_reference_files(tmp_path, file_digests, label):
    """oracle/_ref/sela_ref_bound = the reference's own sela::Encoder / sela::Decoder class declarations and its unmodified
    file classes (src/file/*.cpp), with processFrames() replaced by one call into libsela_hip.so (oracle/binding/bound.cpp,
    built by `make -C oracle bound` in the build container; src/lpc, src/rice, src/frame are not linked).  Its -e / -d
    write the very files the unmodified reference wrote (tests/golden/file_digests.json)."""
    import subprocess

    from sela_amd.synth import synth_pcm
    from gpu_common import _sha_file, _write_wav

    if not os.path.exists(BOUND):
        pytest.fail("oracle/_ref/sela_ref_bound is missing: run `make -C oracle bound` in the build container (it travels to the GPU box)")
    d = file_digests[label]
    pcm = synth_pcm(d["samples_per_channel"], d["channels"], d["track"])
    wav, sela, back = tmp_path / "in.wav", tmp_path / "out.sela", tmp_path / "back.wav"
    _write_wav(wav, pcm, d["sample_rate"])
    assert _sha_file(wav) == d["wav_sha256"], "input drifted"
    r = subprocess.run([BOUND, "-e", str(wav), str(sela)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(sela) == d["sela_bytes"] and _sha_file(sela) == d["sela_sha256"]
    r = subprocess.run([BOUND, "-d", str(sela), str(back)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(back) == d["decoded_wav_bytes"] and _sha_file(back) == d["decoded_wav_sha256"]


@pytest.mark.parametrize("label", ["config0_mono_10s", "config1_stereo_3min", "stereo_48k_tail", "three_channel_96k", "shorter_than_a_frame"])
def test_reference_main_compiled_against_this_host_writes_the_reference_files(tmp_path, file_digests, label):
    """The other direction of the boundary: oracle/_ref/sela_ref_main_on_host = the reference's UNCHANGED src/main.cpp
    (/root/reference/src/main.cpp:29-51: `sela::Encoder encoder = sela::Encoder(inputFile); file::SelaFile selaFile =
    encoder.process(); selaFile.writeToFile(outputFile);` ...) compiled against THIS repo's host classes through the
    reference's own header paths (host/compat/include, `make -C oracle main_on_host`).  Its -e / -d write the reference's
    files; its -p plays through sela::Player's default sink (the packets, as they are, on standard output)."""
    import subprocess

    from sela_amd.synth import synth_pcm
    from gpu_common import _sha_file, _write_wav

    if not os.path.exists(MAIN_ON_HOST):
        pytest.fail("oracle/_ref/sela_ref_main_on_host is missing: run `make -C oracle main_on_host` in the build container")
    d = file_digests[label]
    pcm = synth_pcm(d["samples_per_channel"], d["channels"], d["track"])
    wav, sela, back = tmp_path / "in.wav", tmp_path / "out.sela", tmp_path / "back.wav"
    _write_wav(wav, pcm, d["sample_rate"])
    r = subprocess.run([MAIN_ON_HOST, "-e", str(wav), str(sela)], capture_output=True, text=True)
    assert r.returncode == 0 and "Encoding: " in r.stdout, (r.stdout, r.stderr)
    assert os.path.getsize(sela) == d["sela_bytes"] and _sha_file(sela) == d["sela_sha256"]
    r = subprocess.run([MAIN_ON_HOST, "-d", str(sela), str(back)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(back) == d["decoded_wav_bytes"] and _sha_file(back) == d["decoded_wav_sha256"]
    if label == "stereo_48k_tail":  # the player's feed: banner lines, then every decoded frame's interleaved int16 bytes
        r = subprocess.run([MAIN_ON_HOST, "-p", str(sela)], capture_output=True)
        assert r.returncode == 0, r.stderr
        with open(back, "rb") as f:
            decoded = f.read()[44:]
        assert r.stdout.endswith(decoded) and len(decoded) > 0


def test_a_file_with_frames_of_other_lengths_decodes_like_the_reference(gpu, tmp_path, generic_digests):  # noqa: F811
    """A hand-made .sela file (frames of 2048, 1000, 3000, 777 samples, stereo) through `sela_mi355x -d`, the reference's
    unchanged main.cpp on this host (object path: sela::Decoder::process + WavFile::writeToFile) and the reference's classes
    bound to the library: the WAV file the reference's own decoder writes (digest made by make_golden.py)."""
    import os
    import subprocess

    g = generic_digests["odd_file"]
    blob, pcm = gc.odd_file_bytes(oracle().frame_encode)
    assert hashlib.sha256(blob).hexdigest() == g["sela_sha256"] and len(blob) == g["sela_bytes"]
    sela = tmp_path / "odd.sela"
    sela.write_bytes(blob)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tools = [os.path.join(root, "host", "sela_mi355x")]
    on_host = os.path.join(root, "oracle", "_ref", "sela_ref_main_on_host")
    bound = os.path.join(root, "oracle", "_ref", "sela_ref_bound")
    tools += [t for t in (on_host, bound) if os.path.exists(t)]
    for i, exe in enumerate(tools):
        wav = tmp_path / f"odd{i}.wav"
        out = subprocess.run([exe, "-d", str(sela), str(wav)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        data = wav.read_bytes()
        assert len(data) == g["decoded_wav_bytes"] and hashlib.sha256(data).hexdigest() == g["decoded_wav_sha256"], exe
        assert data[44:] == pcm.tobytes()
