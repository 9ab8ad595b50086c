#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REAL REFERENCE.

Runs only in the build container (needs oracle/_ref/libsela_ref.so, i.e. `make -C oracle ref`,
which compiles the unmodified reference from /root/reference).  The fixtures are pure data:
inputs and the reference's outputs.  Commit the resulting .npz / .json files; the GPU box
and CI only ever read them.

    python tests/golden/make_golden.py [kats] [configs] [files] [album] [generic] [ragged]     (default: all six sections)
"""
import ctypes as C
import hashlib
import json
import math
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle_lib import reference  # noqa: E402
from sela_amd.synth import album_tracks, synth_frames, synth_frames_torch, synth_pcm  # noqa: E402


def edge_blocks():
    """The reference's own test signals (test/lpctests.cpp:16-18, test/frametests.cpp:14-24)
    plus the degenerate blocks of SURVEY.md App. C / App. E."""
    n = 2048
    rng = np.random.default_rng(1)
    i = np.arange(n)
    blocks = {
        "sine_deg": np.array([int(32767 * math.sin(j * (math.pi / 180))) for j in range(n)], np.int32),
        "silence": np.zeros(n, np.int32),
        "const_1234": np.full(n, 1234, np.int32),
        "white_fullscale": rng.integers(-32768, 32768, n).astype(np.int32),
        "square_p64": np.where((i // 32) % 2 == 0, 32767, -32768).astype(np.int32),
        "impulse": np.concatenate([[32767], np.zeros(n - 1)]).astype(np.int32),
        "ramp": (i - 1024).astype(np.int32),
        "alternating": np.where(i % 2 == 0, 20000, -20000).astype(np.int32),
        "min_value": np.full(n, -32768, np.int32),
        "white_small": rng.integers(-3, 4, n).astype(np.int32),
        "diff_extreme": (np.where(i % 3 == 0, 32767, -32768) - np.where(i % 5 == 0, -32768, 32767)).astype(np.int32),
    }
    return blocks


def write_wav(path, pcm, rate):
    """Canonical 44-byte-header WAV (what the reference's own writer emits, src/file/wav_file.cpp:222-242)."""
    data = np.ascontiguousarray(pcm, dtype="<i2").tobytes()
    ch = pcm.shape[1]
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IhHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest(), os.path.getsize(path)


def file_digests(ref):
    """FILE-level pins: the reference's own sela::Encoder + SelaFile::writeToFile and sela::Decoder +
    WavFile::writeToFile (ref_encode_file / ref_decode_file in oracle/ref_shim.cpp) on real WAV files --
    whole .sela and decoded .wav files, headers and dropped tails included."""
    lib = ref.lib
    for fn in (lib.ref_encode_file, lib.ref_decode_file):
        fn.argtypes = [C.c_char_p, C.c_char_p]
        fn.restype = C.c_int
    out = {}
    cases = [("config0_mono_10s", 441000, 1, 44100, 0), ("config1_stereo_3min", 7938000, 2, 44100, 0),
             ("stereo_48k_tail", 5 * 2048 + 777, 2, 48000, 17), ("three_channel_96k", 3 * 2048 + 5, 3, 96000, 5),
             ("shorter_than_a_frame", 100, 2, 44100, 9)]
    with tempfile.TemporaryDirectory() as tmp:
        for label, n, ch, rate, track in cases:
            wav, sela, back = (os.path.join(tmp, label + e) for e in (".wav", ".sela", ".back.wav"))
            write_wav(wav, synth_pcm(n, ch, track), rate)
            assert lib.ref_encode_file(wav.encode(), sela.encode()) == 0
            assert lib.ref_decode_file(sela.encode(), back.encode()) == 0
            (wsha, wsize), (ssha, ssize), (bsha, bsize) = sha_file(wav), sha_file(sela), sha_file(back)
            out[label] = {"samples_per_channel": n, "channels": ch, "sample_rate": rate, "track": track,
                          "wav_sha256": wsha, "wav_bytes": wsize, "sela_sha256": ssha, "sela_bytes": ssize,
                          "decoded_wav_sha256": bsha, "decoded_wav_bytes": bsize}
            print(label, out[label])
    return out


def album_digests(ref, threads=8):
    """BASELINE.json configs[3]: the 100-track album (sela_amd.synth.album_tracks), every track encoded and
    decoded by the reference; per track the SHA-256 of the whole .sela FILE (15-byte header + frames) and
    of the decoded PCM, plus one digest over all of them."""
    tracks = []
    total = hashlib.sha256()
    sizes = hashlib.sha256()  # every frame's size (uint64 LE) in job order: the layout the ranks agree on
    for track, rate, frames in album_tracks():
        pcm = synth_frames_torch(frames, 2, track).numpy()
        blob, offs, _ = ref.encode_frames(pcm, threads=threads)
        dec, _ = ref.decode_frames(blob, offs, 2, threads=threads)
        header = b"SeLa" + struct.pack("<IHBI", rate, 16, 2, frames)
        sela_sha = hashlib.sha256(header + blob.tobytes()).hexdigest()
        dec_sha = hashlib.sha256(dec.tobytes()).hexdigest()
        total.update(bytes.fromhex(sela_sha))
        sizes.update(np.diff(offs.astype(np.uint64)).astype("<u8").tobytes())
        tracks.append({"track": track, "sample_rate": rate, "n_frames": frames, "sela_bytes": 15 + int(len(blob)),
                       "sela_sha256": sela_sha, "decoded_sha256": dec_sha,
                       "lossy_frames": int((dec != pcm).reshape(frames, -1).any(axis=1).sum())})
        print(tracks[-1], flush=True)
    return {"n_tracks": len(tracks), "n_frames": sum(t["n_frames"] for t in tracks), "channels": 2,
            "sha256_of_track_sela_sha256s": total.hexdigest(), "frame_sizes_sha256": sizes.hexdigest(), "tracks": tracks}


def main():
    ref = reference()
    assert ref is not None, "build oracle/_ref first: make -C oracle ref"
    sections = set(sys.argv[1:]) or {"kats", "configs", "files", "album", "generic", "ragged"}
    if "generic" in sections:
        generic(ref)
    if "ragged" in sections:
        ragged(ref)
    if "files" in sections:
        with open(os.path.join(HERE, "file_digests.json"), "w") as f:
            json.dump(file_digests(ref), f, indent=1, sort_keys=True)
    if "album" in sections:
        with open(os.path.join(HERE, "album_digests.json"), "w") as f:
            json.dump(album_digests(ref), f, indent=1, sort_keys=True)
    if "kats" in sections:
        kats(ref)
    if "configs" in sections:
        config_digests(ref)


def ragged(ref):
    """Frames whose channels differ in length through the reference's frame::FrameEncoder (oracle/ref_shim.cpp
    ref_frame_encode_ragged) and back through its frame::FrameDecoder: digests of the inputs (regenerated by
    tests/generic_cases.py), the frame bytes (whole, for the short ones) and the decoded channels -> ragged.json, ragged_kats.npz."""
    import generic_cases as gc

    table, arrays = {}, {}
    for label, chans in gc.ragged_cases():
        blob = ref.frame_encode_ragged(chans)
        dec, used = ref.frame_decode_i32(blob, len(chans))
        assert used == len(blob)
        table[label] = {"lengths": [int(len(c)) for c in chans], "input_sha256": gc.sha_channels(chans), "frame_bytes": len(blob),
                        "frame_sha256": hashlib.sha256(blob).hexdigest(), "decoded_sha256": gc.sha_channels(dec),
                        "subframe_types": [int(s[1]) for s in gc.subframes_of(blob, len(chans))],
                        "lossless": bool(all(np.array_equal(a, b) for a, b in zip(dec, chans)))}
        if len(blob) < 20000:
            arrays[f"{label}/bytes"] = np.frombuffer(blob, np.uint8)
        print(label, table[label], flush=True)
    with open(os.path.join(HERE, "ragged.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "ragged_kats.npz"), **arrays)


def generic(ref):
    """Frames of any length and 17-bit samples through the reference's frame classes (frame::FrameEncoder on a data::WavFrame,
    frame::FrameDecoder on the result: oracle/ref_shim.cpp ref_frame_encode_i32 / ref_frame_decode_i32): per case the digests of
    the input (regenerated by tests/generic_cases.py), of the frame bytes and of the decoded channels; the bytes themselves for
    the short ones; and crafted frames (spliced subframes) with the reference's decoded channels."""
    import generic_cases as gc

    table, arrays = {}, {}
    for label, n, kind, wide in gc.all_cases():
        x = gc.case_input(n, kind, wide)
        blob = ref.frame_encode_i32(x)
        dec, used = ref.frame_decode_i32(blob, x.shape[0])
        assert used == len(blob)
        table[label] = {"n": n, "kind": kind, "wide": wide, "channels": int(x.shape[0]), "input_sha256": gc.sha(x),
                        "frame_bytes": len(blob), "frame_sha256": hashlib.sha256(blob).hexdigest(), "decoded_sha256": gc.sha_channels(dec),
                        "lossless": bool(all(np.array_equal(a, b) for a, b in zip(dec, x)))}
        if n <= 2049:
            arrays[f"{label}/bytes"] = np.frombuffer(blob, np.uint8)
        print(label, table[label], flush=True)

    # ---- crafted frames --------------------------------------------------------------------------------------------------------
    a = ref.frame_encode_i32(gc.case_input(300, "three", True))     # three independent subframes of 300
    b = ref.frame_encode_i32(gc.case_input(200, "three", False))    # ... and of 200
    sa, sb = gc.subframes_of(a, 3), gc.subframes_of(b, 3)
    crafted = {
        # channels of different lengths: every subframe keeps its own samplesPerChannel (frame_decoder.cpp:24-25)
        "mixed_lengths": (gc.join_frame([sa[0], gc.retag(sb[1], 1, 0, 1), sa[2]]), 3),
        # a chain of dependent subframes, resolved in subframe order (:40-69): ch1 = ch0 - d1, ch2 = ch1 - d2
        "dependent_chain": (gc.join_frame([sa[0], gc.retag(sa[1], 1, 1, 0), gc.retag(sa[2], 2, 1, 1)]), 3),
        # the dependent subframe comes FIRST in the frame: independent ones are still decoded first (:17-37)
        "dependent_first": (gc.join_frame([gc.retag(sa[1], 1, 1, 0), gc.retag(sa[0], 0, 0, 0), sa[2]]), 3),
        # two subframes name the same channel: the later one wins; channel 2 stays empty
        "same_channel_twice": (gc.join_frame([sa[0], gc.retag(sa[1], 0, 0, 0), gc.retag(sb[2], 1, 0, 1)]), 3),
        # a parent longer than the difference (only the difference's length is produced)
        "long_parent": (gc.join_frame([sa[0], gc.retag(sb[1], 1, 1, 0), sa[2]]), 3),
        # a subframe type that is neither 0 nor 1 is skipped by both passes
        "unknown_type": (gc.join_frame([sa[0], gc.retag(sa[1], 1, 7, 0), sa[2]]), 3),
    }
    names = []
    for name, (blob, ch) in crafted.items():
        dec, used = ref.frame_decode_i32(blob, ch)
        assert used == len(blob)
        names.append(name)
        arrays[f"crafted/{name}/bytes"] = np.frombuffer(blob, np.uint8)
        arrays[f"crafted/{name}/channels"] = np.int32(ch)
        for c in range(ch):
            arrays[f"crafted/{name}/decoded{c}"] = dec[c]
        print("crafted", name, [len(d) for d in dec], flush=True)
    arrays["crafted_names"] = np.array(names)

    # ---- a hand-made file with frames of different lengths through the reference's own decoder + WAV writer -------------------
    lib = ref.lib
    lib.ref_decode_file.argtypes = [C.c_char_p, C.c_char_p]
    lib.ref_decode_file.restype = C.c_int
    blob, pcm = gc.odd_file_bytes(ref.frame_encode)
    with tempfile.TemporaryDirectory() as tmp:
        sela, wav = os.path.join(tmp, "odd.sela"), os.path.join(tmp, "odd.wav")
        with open(sela, "wb") as f:
            f.write(blob)
        assert lib.ref_decode_file(sela.encode(), wav.encode()) == 0
        wsha, wsize = sha_file(wav)
        with open(wav, "rb") as f:
            assert f.read()[44:] == pcm.tobytes()
    table["odd_file"] = {"sela_sha256": hashlib.sha256(blob).hexdigest(), "sela_bytes": len(blob), "decoded_wav_sha256": wsha,
                         "decoded_wav_bytes": wsize, "frame_lengths": list(gc.ODD_FILE_LENGTHS)}
    print("odd_file", table["odd_file"], flush=True)
    with open(os.path.join(HERE, "generic.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "generic_kats.npz"), **arrays)


def kats(ref):
    out = {}

    # ---- per-stage KATs on single blocks --------------------------------------------------
    names = []
    for name, s in edge_blocks().items():
        order, q, r = ref.lpc_analyze(s)
        a = ref.lpc_coeffs(order, q)
        ck, cw = ref.rice_encode(q)
        rk, rw = ref.rice_encode(r)
        back = ref.lpc_synth(order, q, r)
        names.append(name)
        out[f"blk/{name}/samples"] = s
        out[f"blk/{name}/order"] = np.int32(order)
        out[f"blk/{name}/q"] = q
        out[f"blk/{name}/a"] = a
        out[f"blk/{name}/residues"] = r
        out[f"blk/{name}/coef_k"] = np.uint32(ck)
        out[f"blk/{name}/coef_words"] = cw
        out[f"blk/{name}/res_k"] = np.uint32(rk)
        out[f"blk/{name}/res_words"] = rw
        out[f"blk/{name}/synth"] = back
    out["blk_names"] = np.array(names)

    # ---- Rice KATs (SURVEY.md App. C + test/ricetests.cpp-style values) ---------------------
    rng = np.random.default_rng(7)
    rice_cases = {
        "small": np.array([0, -1, 1, -2, 2, 100, -100, 5], np.int32),
        "ricetest_like": (200 + rng.integers(0, 201, 100)).astype(np.int32),
        "zeros": np.zeros(64, np.int32),
        "single": np.array([-7], np.int32),
        "long_unary": np.array([0, 0, 5000, 0, -4000, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], np.int32),
        "wide": rng.integers(-(1 << 20), 1 << 20, 257).astype(np.int32),
    }
    out["rice_names"] = np.array(list(rice_cases))
    for name, v in rice_cases.items():
        k, w = ref.rice_encode(v)
        assert np.array_equal(ref.rice_decode(w, len(v), k), v)
        out[f"rice/{name}/values"] = v
        out[f"rice/{name}/k"] = np.uint32(k)
        out[f"rice/{name}/words"] = w

    # ---- frame KATs: on-disk bytes of whole frames ----------------------------------------------
    sine = edge_blocks()["sine_deg"].astype(np.int16)
    frames = {
        "stereo_same_sine": np.stack([sine, sine], axis=1),  # test/frametests.cpp:40-70
        "mono_sine": sine[:, None],
        "stereo_synth0": synth_frames(9, 2, 0)[8],   # w == 2 region
        "stereo_synth_diff": synth_frames(20, 2, 0)[17],  # fully common noise -> difference coding
        "stereo_synth_indep": synth_frames(2, 2, 0)[1],
        "mono_synth": synth_frames(3, 1, 3)[2],
        "three_channel": synth_frames(2, 3, 5)[1],
        "stereo_silence": np.zeros((2048, 2), np.int16),
    }
    out["frame_names"] = np.array(list(frames))
    for name, pcm in frames.items():
        blob = ref.frame_encode(pcm)
        dec, used = ref.frame_decode(blob, pcm.shape[1])
        assert used == len(blob)
        out[f"frame/{name}/pcm"] = pcm.astype(np.int16)
        out[f"frame/{name}/bytes"] = np.frombuffer(blob, np.uint8)
        out[f"frame/{name}/decoded"] = dec

    np.savez_compressed(os.path.join(HERE, "kats.npz"), **out)


def config_digests(ref):
    # ---- whole-config digests (BASELINE.json configs; inputs are regenerated from synth) -------
    digests = {}
    for label, nf, ch, track in [("config0_mono_10s", 215, 1, 0), ("config1_stereo_3min", 3875, 2, 0),
                                 ("config2_1000_frames", 1000, 2, 1)]:
        pcm = synth_frames(nf, ch, track)
        blob, offs, _ = ref.encode_frames(pcm, threads=8)
        dec, _ = ref.decode_frames(blob, offs, ch, threads=8)
        digests[label] = {
            "n_frames": nf, "channels": ch, "track": track,
            "pcm_sha256": hashlib.sha256(pcm.tobytes()).hexdigest(),
            "frames_blob_sha256": hashlib.sha256(blob.tobytes()).hexdigest(),
            "frames_blob_bytes": int(len(blob)),
            "offsets_sha256": hashlib.sha256(offs.astype("<u8").tobytes()).hexdigest(),
            "decoded_sha256": hashlib.sha256(dec.tobytes()).hexdigest(),
            "lossless": bool(np.array_equal(dec, pcm)),
        }
        print(label, digests[label])
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
