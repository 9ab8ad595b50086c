"""The C++ host (host/): container code on the CPU, the CLI and the frame classes on the GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sela_amd.synth import synth_pcm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")


def _build():
    if not os.path.exists(os.path.join(ROOT, "sela_amd", "libsela_hip.so")):
        import __graft_entry__ as g

        g.build_hip_library()
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)


def _write_wav(path, pcm, rate=44100):
    data = np.ascontiguousarray(pcm, dtype="<i2").tobytes()
    ch = pcm.shape[1]
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IhHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def test_container_selftest(tmp_path):
    """WAV / .sela parse + serialise round trips and the reference's error messages (no GPU)."""
    _build()
    out = subprocess.run([os.path.join(HOST, "host_selftest"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_cli_without_gpu_fails_loudly(tmp_path):
    """No device -> exit code 1 and an error message; never a silent CPU encode."""
    from sela_amd import capi

    _build()
    if capi.lib().sela_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    wav = tmp_path / "in.wav"
    _write_wav(wav, synth_pcm(4096, 2, 0))
    out = subprocess.run([os.path.join(HOST, "sela_mi355x"), "-e", str(wav), str(tmp_path / "out.sela")], capture_output=True, text=True)
    assert out.returncode == 1
    assert "no HIP device" in out.stderr


@pytest.mark.gpu
def test_frame_classes_on_gpu(tmp_path):
    """frame::FrameEncoder / FrameDecoder: the reference's test/frametests.cpp through the host classes."""
    _build()
    out = subprocess.run([os.path.join(HOST, "host_selftest"), str(tmp_path), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("channels,seconds", [(2, 3.3), (1, 2.1)])
def test_cli_files_match_reference_bytes(tmp_path, channels, seconds):
    """`sela_mi355x -e` writes the byte-identical .sela the reference writes (header + frame stream,
    tail dropped), `-d` restores the WAV prefix bit-exactly."""
    from oracle_lib import oracle

    _build()
    n = int(44100 * seconds)
    pcm = synth_pcm(n, channels, 17)
    wav, sela, back = tmp_path / "in.wav", tmp_path / "out.sela", tmp_path / "back.wav"
    _write_wav(wav, pcm)
    cli = os.path.join(HOST, "sela_mi355x")
    r = subprocess.run([cli, "-e", str(wav), str(sela)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    frames = n // 2048
    blob, offs, _ = oracle().encode_frames(pcm[: frames * 2048].reshape(frames, 2048, channels), threads=4)
    expect = b"SeLa" + struct.pack("<IHBI", 44100, 16, channels, frames) + blob.tobytes()
    assert open(sela, "rb").read() == expect
    r = subprocess.run([cli, "-d", str(sela), str(back)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(back, "rb").read()
    assert raw[:4] == b"RIFF" and raw[36:40] == b"data"
    got = np.frombuffer(raw[44:], dtype="<i2").reshape(-1, channels)
    assert np.array_equal(got, pcm[: frames * 2048])  # the tail is dropped


@pytest.mark.gpu
def test_cli_batch_of_files_equals_file_by_file(tmp_path):
    """`-E` / `-D` push many files through ONE GPU batch per channel count (BASELINE.json configs[3]);
    every output must be the byte-identical file the single-file verbs write."""
    _build()
    cli = os.path.join(HOST, "sela_mi355x")
    specs = [("a", 2, 44100, 5 * 2048 + 777), ("b", 2, 48000, 3 * 2048), ("c", 1, 96000, 2 * 2048 + 5), ("d", 2, 44100, 100), ("e", 1, 44100, 4 * 2048)]
    wavs = []
    for name, ch, rate, n in specs:
        p = tmp_path / f"{name}.wav"
        _write_wav(p, synth_pcm(n, ch, 60 + len(wavs)), rate)
        wavs.append(p)
    single, batch, back1, backn = (tmp_path / d for d in ("single", "batch", "back1", "backn"))
    for d in (single, batch, back1, backn):
        d.mkdir()
    for w in wavs:
        r = subprocess.run([cli, "-e", str(w), str(single / (w.stem + ".sela"))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([cli, "-E", str(batch)] + [str(w) for w in wavs], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for w in wavs:
        assert (batch / (w.stem + ".sela")).read_bytes() == (single / (w.stem + ".sela")).read_bytes(), w.stem
    selas = [batch / (w.stem + ".sela") for w in wavs]
    for s in selas:
        r = subprocess.run([cli, "-d", str(s), str(back1 / (s.stem + ".wav"))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([cli, "-D", str(backn)] + [str(s) for s in selas], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for s in selas:
        assert (backn / (s.stem + ".wav")).read_bytes() == (back1 / (s.stem + ".wav")).read_bytes(), s.stem


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [2, 1])
def test_cli_play_hands_out_the_decoded_frames_in_order(tmp_path, channels):
    """`-p`: the packets the reference's player would give to libao (src/sela/player.cpp:30-62: one interleaved int16 packet
    per frame, in order) are the decoded data chunk cut into frames -- to a file and to standard output, and for a stream
    that stops at a frame without its sync word (src/file/sela_file.cpp:54-56) only the frames before it."""
    from oracle_lib import oracle

    _build()
    frames = 70
    pcm = synth_pcm(frames * 2048, channels, 23).reshape(frames, 2048, channels)
    blob, offs, _ = oracle().encode_frames(pcm, threads=4)
    expect, _ = oracle().decode_frames(blob, offs, channels, threads=4)  # what the reference's decoder gives (lossy ties and all)
    sela = tmp_path / "in.sela"
    sela.write_bytes(b"SeLa" + struct.pack("<IHBI", 44100, 16, channels, frames) + blob.tobytes())
    cli = os.path.join(HOST, "sela_mi355x")
    out = tmp_path / "out.pcm"
    r = subprocess.run([cli, "-p", str(sela), str(out)], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == b""
    assert f"{frames} frames".encode() in r.stderr
    assert out.read_bytes() == np.ascontiguousarray(expect, dtype="<i2").tobytes()
    r = subprocess.run([cli, "-p", str(sela)], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == out.read_bytes()  # nothing but samples on standard output
    # a stream broken in frame 41: frames 0..40 are played, like the reference's reader stops there
    raw = bytearray(sela.read_bytes())
    raw[15 + int(offs[41])] ^= 0xFF
    cut = tmp_path / "cut.sela"
    cut.write_bytes(bytes(raw))
    r = subprocess.run([cli, "-p", str(cut)], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == out.read_bytes()[: 41 * 2048 * channels * 2]
