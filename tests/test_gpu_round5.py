"""GPU parity tests of round 5: the any-length / 32-bit route (sela_generic.hip through sela_hip_encode / sela_hip_decode with
samples_per_channel != 2048, sela_hip_encode_i32 / sela_hip_decode_i32, sela_hip_lpc_*_n) against fixtures the unmodified
reference wrote (tests/golden/generic.json, generic_kats.npz) and against the oracle -- bit-exact: frame bytes, offsets,
decoded 32-bit channels."""
import hashlib

import numpy as np
import pytest

import generic_cases as gc
from oracle_lib import oracle
from sela_amd.synth import synth_frames, synth_pcm
from test_gpu_parity import gpu  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _one(offs):
    return np.array([0, offs], np.uint64)


@pytest.mark.parametrize("n", gc.LENGTHS)
def test_generic_frames_against_the_reference_fixtures(gpu, generic_digests, generic_kats, n):  # noqa: F811
    """n in {128, 1000, 2047, 2049, 4096, 65535} x {mono, stereo (difference wins / loses), three channels} x {16-bit, 17-bit}:
    frame bytes and decoded int32 channels equal the reference's (digests; whole bytes for the short ones)."""
    from sela_amd import codec

    o = oracle()
    for label, nn, kind, wide in gc.all_cases():
        if nn != n:
            continue
        g = generic_digests[label]
        x = gc.case_input(n, kind, wide)
        assert gc.sha(x) == g["input_sha256"], label
        frames, offs = codec.encode_i32(x[None])
        blob = frames.tobytes()
        if hashlib.sha256(blob).hexdigest() != g["frame_sha256"]:  # say where, with the oracle's help
            want = o.frame_encode_i32(x)
            first = next((i for i, (a, b) in enumerate(zip(blob, want)) if a != b), min(len(blob), len(want)))
            pytest.fail(f"{label}: frame bytes differ from the reference's at byte {first} ({len(blob)} vs {len(want)} bytes)")
        assert int(offs[1]) == g["frame_bytes"], label
        if f"{label}/bytes" in generic_kats:
            assert blob == generic_kats[f"{label}/bytes"].tobytes(), label
        dec = codec.decode_i32(frames, offs, x.shape[0])[0]
        assert gc.sha_channels(dec) == g["decoded_sha256"], label
        assert all(np.array_equal(a, b) for a, b in zip(dec, x)) == g["lossless"], label
        if not wide:  # the int16 entry points: the same frame, the same samples
            pcm = np.ascontiguousarray(x.T.astype(np.int16))[None]
            f16, o16 = codec.encode_host(pcm)
            assert f16.tobytes() == blob and int(o16[1]) == len(blob), label
            back = codec.decode_host(frames, offs, x.shape[0])
            assert back.shape == (n, x.shape[0]) and np.array_equal(back, pcm[0]), label
        else:  # 17-bit samples through the 16-bit writer: truncated like src/file/wav_file.cpp:248-251
            back = codec.decode_host(frames, offs, x.shape[0])
            assert np.array_equal(back, x.T.astype(np.uint32).astype(np.uint16).view(np.int16)), label


def test_crafted_frames_decode_like_the_reference(gpu, generic_kats):  # noqa: F811
    """Subframes spliced by hand: channels of different lengths, a chain of dependent subframes, a dependent subframe ahead of
    its parent, one channel named twice, a long parent, an unknown subframe type -- FrameDecoder's answer, channel by channel."""
    from sela_amd import capi, codec

    for name in generic_kats["crafted_names"]:
        blob = generic_kats[f"crafted/{name}/bytes"]
        ch = int(generic_kats[f"crafted/{name}/channels"])
        dec = codec.decode_i32(blob, _one(len(blob)), ch)[0]
        lengths = []
        for c in range(ch):
            want = generic_kats[f"crafted/{name}/decoded{c}"]
            assert np.array_equal(dec[c], want), (name, c)
            lengths.append(len(want))
        if len(set(lengths)) > 1:  # no interleaved PCM exists for channels of different lengths
            with pytest.raises(capi.SelaHipError) as err:
                codec.decode_host(blob, _one(len(blob)), ch)
            assert err.value.code == -5, name
        else:
            back = codec.decode_host(blob, _one(len(blob)), ch)
            want = np.stack([generic_kats[f"crafted/{name}/decoded{c}"] for c in range(ch)], axis=1)
            assert np.array_equal(back, want.astype(np.uint32).astype(np.uint16).view(np.int16)), name


def test_batches_of_odd_frames_and_mixed_streams(gpu):  # noqa: F811
    """Many frames per call (offsets, chunking), and ONE stream whose frames have different lengths, 2048 among them."""
    from sela_amd import codec

    o = oracle()
    blobs, pcms = [], []
    for n, ch, nf, track in ((1000, 2, 37, 3), (2047, 1, 9, 4), (2049, 3, 5, 5), (4096, 2, 11, 6)):
        pcm = synth_pcm(n * nf, ch, track).reshape(nf, n, ch)
        frames, offs = codec.encode_host(pcm)
        want = [o.frame_encode(pcm[f]) for f in range(nf)]
        assert frames.tobytes() == b"".join(want), (n, ch)
        assert offs.tolist() == np.concatenate([[0], np.cumsum([len(w) for w in want])]).tolist(), (n, ch)
        back = codec.decode_host(frames, offs, ch)
        assert np.array_equal(back, pcm.reshape(-1, ch)), (n, ch)
        planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
        f32, o32 = codec.encode_i32(planar)
        assert f32.tobytes() == frames.tobytes() and np.array_equal(o32, offs)
        if ch == 2:
            blobs += want
            pcms += [pcm[f] for f in range(nf)]
    # a stereo stream of 1000-, 4096- and 2048-sample frames, interleaved
    std = synth_frames(6, 2, 9)
    blobs += [o.frame_encode(std[f]) for f in range(6)]
    pcms += [std[f] for f in range(6)]
    order = np.random.default_rng(0).permutation(len(blobs))
    stream = b"".join(blobs[i] for i in order)
    offs = np.concatenate([[0], np.cumsum([len(blobs[i]) for i in order])]).astype(np.uint64)
    so, largest = codec.index_samples(np.frombuffer(stream, np.uint8), offs, 2)
    assert largest == 4096 and so.tolist() == np.concatenate([[0], np.cumsum([len(pcms[i]) for i in order])]).tolist()
    back = codec.decode_host(np.frombuffer(stream, np.uint8), offs, 2)
    assert np.array_equal(back, np.concatenate([pcms[i] for i in order]))
    dec = codec.decode_i32(np.frombuffer(stream, np.uint8), offs, 2)
    for j, i in enumerate(order):
        assert np.array_equal(np.stack(dec[j], axis=1), pcms[i].astype(np.int32)), j


def test_2048_through_the_generic_route_equals_the_fast_kernels(gpu, kats):  # noqa: F811
    """The same frames through both routes: bytes and samples identical (stereo with both decisions, mono, the KAT blocks)."""
    from sela_amd import codec

    for pcm in (synth_frames(24, 2, 0), synth_frames(7, 1, 2), synth_frames(5, 3, 4)):
        fast, fo = codec.encode_host(pcm)
        planar = np.ascontiguousarray(pcm.transpose(0, 2, 1)).astype(np.int32)
        gen, go = codec.encode_i32(planar)
        assert gen.tobytes() == fast.tobytes() and np.array_equal(go, fo)
        dec = codec.decode_i32(fast, fo, pcm.shape[2])
        for f in range(pcm.shape[0]):
            assert np.array_equal(np.stack(dec[f], axis=1), pcm[f].astype(np.int32)), f
    names = [str(n) for n in kats["blk_names"]]
    blocks = np.stack([kats[f"blk/{n}/samples"] for n in names]).astype(np.int32)  # (diff_extreme is 17-bit)
    gen, go = codec.encode_i32(blocks[:, None, :])
    o = oracle()
    assert gen.tobytes() == b"".join(o.frame_encode_i32(b[None]) for b in blocks)


def test_what_the_reference_cannot_answer_is_refused(gpu):  # noqa: F811
    """A block not longer than its own order (the reference reads past its vector), lengths the u16 field cannot say,
    a stride too small, residues beyond the int32 zig-zag."""
    from sela_amd import capi, codec

    rng = np.random.default_rng(2)
    noise = rng.integers(-20000, 20000, (1, 1, 40)).astype(np.int32)  # white noise: the order comes out above 40
    assert oracle().lpc_analyze(noise[0, 0])[0] >= 40
    with pytest.raises(capi.SelaHipError) as err:
        codec.encode_i32(noise)
    assert err.value.code == -6
    for bad in (0, 65536):
        lib = capi.lib()
        buf = np.zeros(16, np.int16)
        out = np.zeros(4096, np.uint8)
        offs = np.zeros(2, np.uint64)
        assert lib.sela_hip_encode(buf.ctypes.data, 1, 1, bad, out.ctypes.data, out.nbytes, offs.ctypes.data) == -2
    x = gc.case_input(1000, "mono", False)
    frames, offs = codec.encode_i32(x[None])
    with pytest.raises(capi.SelaHipError) as err:
        codec.decode_i32(frames, offs, 1, stride=999)
    assert err.value.code == -4
    wild = (rng.integers(-(1 << 31), 1 << 31, (1, 1, 500))).astype(np.int32)  # full-range int32: residues overflow the zig-zag
    with pytest.raises(capi.SelaHipError) as err:
        codec.encode_i32(wild)
    assert err.value.code == -6
    # a short frame that IS longer than its order: a constant block has order 1
    flat = np.full((1, 1, 2), 5, np.int32)
    frames, offs = codec.encode_i32(flat)
    assert frames.tobytes() == oracle().frame_encode_i32(flat[0])


def test_lpc_stages_of_any_length(gpu):  # noqa: F811
    """lpc::ResidueGenerator / SampleGenerator on vectors of any length and any 32-bit value, against the oracle."""
    from sela_amd import codec

    o = oracle()
    rng = np.random.default_rng(8)
    for n, amp in ((101, 3000), (500, 32767), (2048, 1 << 20), (4096, 65535), (10000, 1 << 22)):
        t = np.arange(n)
        blocks = np.stack([np.clip(np.round(amp * 0.6 * np.sin(t * (0.01 + 0.003 * b)) + rng.normal(0, amp * 0.02, n)), -amp, amp) for b in range(6)]).astype(np.int32)
        order, q, res = codec.lpc_encode_n(blocks)
        for b in range(len(blocks)):
            wo, wq, wr = o.lpc_analyze(blocks[b])
            assert order[b] == wo and np.array_equal(q[b, :wo], wq) and not q[b, wo:].any() and np.array_equal(res[b], wr), (n, b)
        back, coefs = codec.lpc_decode_n(order, q, res, want_coefficients=True)
        for b in range(len(blocks)):
            assert np.array_equal(back[b], o.lpc_synth(int(order[b]), q[b, : order[b]], res[b])), (n, b)
            assert np.array_equal(coefs[b, : order[b] + 1], o.lpc_coeffs(int(order[b]), q[b, : order[b]])), (n, b)
    # the 2048 entry with samples beyond 17 bits goes the same way, and zeroes q beyond the order
    big = (rng.integers(-(1 << 19), 1 << 19, (3, 2048))).astype(np.int32)
    order, q, res = codec.lpc_encode(big)
    for b in range(3):
        wo, wq, wr = o.lpc_analyze(big[b])
        assert order[b] == wo and np.array_equal(q[b, :wo], wq) and not q[b, wo:].any() and np.array_equal(res[b], wr)


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


def test_rice_stage_cuts_a_stream_beyond_2_24_bits_like_the_reference(gpu):  # noqa: F811
    """sela_hip_rice_encode on a stream whose float-rounded word count is one short: the reference's k, count and
    words -- nothing written past them (the stream behind it in the same call is intact)."""
    from sela_amd import codec

    v, k, bits = gc.long_rice_stream()
    tail = np.array([0, -1, 1, -2, 2, 100, -100, 5], np.int32)
    (k0, w0), (k1, w1) = codec.rice_encode([v, tail])
    ko, wo = oracle().rice_encode(v)
    assert k0 == ko == k and len(w0) == (bits + 31) // 32 - 1 and np.array_equal(w0, wo)
    assert k1 == 5 and [hex(x) for x in w1] == ["0xc8c10800", "0x538fc4f"]


def test_wide_differential_corpus_against_the_reference(gpu):  # noqa: F811
    """33,400 stereo frames = 100,200 analysed blocks (tests/corpus.py: AR(2..32) noise at many levels, |k| hovering at 0.05,
    clipped and faded tones, DC steps, silence <-> full scale inside a block, 17-bit differences, loud and smooth polyphony)
    through all three encode kernels and the decoder against the UNMODIFIED reference (oracle/_ref/libsela_ref.so on all host
    threads; the oracle where that library is absent): frame bytes, offsets, decoded PCM.  And which form of the residue
    filter every block took, unforced, read back from the product kernels' per-block records: equal to the documented rule
    evaluated on the oracle's predictor for a sample of blocks; one pass and two passes both occur (the plain loop does not:
    no 16-bit input was found that gets a predictor past 2^39 through the reference's quantiser -- it is reached by the
    forced-form tests of round 4 and by the trace builds)."""
    import ctypes as C
    import os
    import time

    import corpus
    from oracle_lib import reference
    from sela_amd import capi, codec

    t0 = time.time()
    pcm = corpus.build()
    n = pcm.shape[0]
    ref = reference() or oracle()
    threads = os.cpu_count() or 8
    want, want_offs, _ = ref.encode_frames(pcm, threads=threads)
    want_dec, _ = ref.decode_frames(want, want_offs, 2, threads=threads)
    t1 = time.time()
    lib = capi.lib()
    enc = codec.Encoder(n, 2)
    enc.frames = gpu.empty(int(lib.sela_hip_encode_bound_bytes(n, 2)), dtype=gpu.uint8, device="cuda")
    enc.capacity = enc.frames.numel()
    d_pcm = gpu.from_numpy(pcm).cuda()
    o = oracle()
    seen = np.zeros(3, np.int64)
    try:
        for teams in (-1, 0, 16, 8):
            lib.sela_hip_debug_encode_teams(teams)
            out = enc.encode(d_pcm)
            gpu.cuda.synchronize()
            frames, offs = out.to_host()
            assert np.array_equal(offs, want_offs), teams
            assert np.array_equal(frames, want), teams
            counts = (C.c_uint32 * 3)()
            forms = np.zeros(n * 3, np.uint8)
            assert lib.sela_hip_debug_block_forms(enc.workspace.data_ptr(), n, 2, counts, forms.ctypes.data) == 0
            seen += np.array(list(counts))
            assert counts[0] > 0 and counts[1] > 0 and sum(counts) == 3 * n, (teams, list(counts))
            if teams == -1:
                picked = list(counts)
                rng = np.random.default_rng(5)
                sample = set(rng.integers(0, n, 400).tolist()) | set((np.nonzero(forms.reshape(n, 3).any(axis=1))[0][:200]).tolist())
                for f in sorted(sample):
                    l, r = pcm[f, :, 0].astype(np.int32), pcm[f, :, 1].astype(np.int32)
                    for sig, s in enumerate((l, r, l - r)):
                        order, q = o.lpc_analyze(s)[:2]
                        a = o.lpc_coeffs(order, q)
                        assert forms[3 * f + sig] == corpus.expected_form(a, order, s), (f, sig)
    finally:
        lib.sela_hip_debug_encode_teams(-1)
    dec = codec.Decoder(n, 2)
    back = dec.decode(out.frames, out.offsets, n)
    gpu.cuda.synchronize()
    dec.check()
    assert np.array_equal(back.cpu().numpy(), want_dec)
    lossy = int((want_dec != pcm).reshape(n, -1).any(axis=1).sum())
    print(f"\ncorpus: {n} stereo frames, {3 * n} blocks; reference {'libsela_ref.so' if ref.is_reference else 'oracle'} on {threads} threads "
          f"{t1 - t0:.1f} s (with generation); forms by the library's own kernel choice (one pass, two passes, plain) = {picked}; "
          f"frames the reference's own decoder does not return exactly: {lossy}; whole test {time.time() - t0:.1f} s")


def _fold(values):
    """hash_term of sela_encode.hip over an array of doubles, position-keyed, XOR-ed."""
    v = np.asarray(values, dtype=np.float64).copy()
    v[np.isnan(v)] = np.float64("nan")
    x = v.view(np.uint64).copy()
    x[np.isnan(v)] = np.uint64(0x7FF8000000000000)
    with np.errstate(over="ignore"):
        x ^= np.uint64(0x9E3779B97F4A7C15) * (np.arange(len(x), dtype=np.uint64) + np.uint64(1))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return int(np.bitwise_xor.reduce(x))


@pytest.mark.parametrize("teams", [16, 8, 0], ids=["k_encode_teams<.,16>", "k_encode_teams<.,8>", "k_encode_blocks<.,false>"])
def test_fp64_intermediates_of_the_product_kernels_by_hash(gpu, kats, teams):  # noqa: F811
    """The normalised autocorrelation ac[0..100] and the reflection coefficients k[0..99] of the kernels that are TIMED --
    k_encode_teams<0,16>, <0,8>, k_encode_blocks<0,false> plus the few instructions that fold them (their kMode 3
    instantiations; round 4 checked these doubles on the trace builds only) -- as two 64-bit hashes per block against the
    oracle's trace folded the same way: the KAT blocks, the corner blocks (NaN paths), stereo and three-channel frames."""
    from sela_amd import capi, codec
    from test_gpu_parity import _kat_block_frames
    from test_gpu_round4 import _hard_blocks

    lib = capi.lib()
    o = oracle()
    _, mono = _kat_block_frames(kats)
    lib.sela_hip_debug_encode_teams(teams)
    lib.sela_hip_debug_encode_hashes(1)
    try:
        for pcm in (mono, _hard_blocks(), synth_frames(27, 2, 3), synth_frames(5, 3, 4)):
            nf, _, ch = pcm.shape
            n_sig = 3 if ch == 2 else ch
            enc = codec.Encoder(nf, ch, with_trace=True)  # (the trace buffer is more than the 16 bytes per block used here)
            enc.trace.zero_()
            out = enc.encode(gpu.from_numpy(np.ascontiguousarray(pcm)).cuda())
            gpu.cuda.synchronize()
            frames, offs = out.to_host()
            want, want_offs, _ = o.encode_frames(pcm, threads=4)
            assert np.array_equal(frames, want) and np.array_equal(offs, want_offs)
            got = enc.trace[: nf * n_sig * 16].cpu().numpy().view(np.uint64).reshape(nf * n_sig, 2)
            for f in range(nf):
                for sig in range(n_sig):
                    s = (pcm[f, :, 0].astype(np.int32) - pcm[f, :, 1]) if (ch == 2 and sig == 2) else pcm[f, :, sig].astype(np.int32)
                    tr = o.lpc_analyze(s, with_trace=True)[4]
                    assert int(got[f * n_sig + sig, 0]) == _fold(list(tr.ac)), (teams, f, sig, "ac")
                    assert int(got[f * n_sig + sig, 1]) == _fold(list(tr.k)), (teams, f, sig, "k")
    finally:
        lib.sela_hip_debug_encode_hashes(0)
        lib.sela_hip_debug_encode_teams(-1)


def test_wave_priorities_follow_the_neighbours(gpu):  # noqa: F811
    """An encode launch takes the falling wave-priority schedule exactly when no OTHER stream has library work pending
    (sela_capi.hip, Flights): alone on its stream -- every launch; queued behind a 20,000-frame encode that another stream
    still runs -- none; and the bytes do not depend on it (forced on, forced off, by the library)."""
    from sela_amd import capi, codec

    lib = capi.lib()
    pcm = gpu.from_numpy(synth_frames(600, 2, 6)).cuda()
    big = gpu.from_numpy(np.tile(synth_frames(500, 2, 7), (40, 1, 1))).cuda()
    enc_a, enc_b = codec.Encoder(600, 2), codec.Encoder(20000, 2)
    want = None
    for forced in (0x00010203, 0, None):
        if forced is None:
            lib.sela_hip_debug_priorities_adaptive()
        else:
            lib.sela_hip_debug_priorities(forced)
        out = enc_a.encode(pcm)
        gpu.cuda.synchronize()
        frames, offs = out.to_host()
        if want is None:
            want = (frames.copy(), offs.copy())
            ref, ref_offs, _ = oracle().encode_frames(pcm.cpu().numpy(), threads=8)
            assert np.array_equal(frames, ref) and np.array_equal(offs, ref_offs)
        assert np.array_equal(frames, want[0]) and np.array_equal(offs, want[1]), forced
    gpu.cuda.synchronize()
    before = lib.sela_hip_debug_launches_alone()
    for _ in range(5):  # one stream, nothing else in flight: every launch is alone
        enc_a.encode(pcm)
    gpu.cuda.synchronize()
    assert lib.sela_hip_debug_launches_alone() - before == 5
    s1, s2 = gpu.cuda.Stream(), gpu.cuda.Stream()
    gpu.cuda.synchronize()
    before = lib.sela_hip_debug_launches_alone()
    with gpu.cuda.stream(s1):
        enc_b.encode(big)  # (several milliseconds of work: alone when it was queued)
    with gpu.cuda.stream(s2):
        for _ in range(3):
            enc_a.encode(pcm)  # queued while the other stream's launch is pending: a neighbour
    gpu.cuda.synchronize()
    assert lib.sela_hip_debug_launches_alone() - before == 1
    out = enc_a.encode(pcm)
    gpu.cuda.synchronize()
    frames, offs = out.to_host()
    assert np.array_equal(frames, want[0]) and np.array_equal(offs, want[1])


def test_random_shapes_against_the_oracle(gpu):  # noqa: F811
    """300 frames of random shape -- 1 .. 6000 samples, 1 .. 6 channels, amplitudes from 1 to 2^20, silent and constant
    channels, near-copies (difference coding), lengths around the analysis' landmarks (63 .. 65, 100 .. 102, 127 .. 129) --
    through sela_hip_encode_i32 / sela_hip_decode_i32 against the oracle; a frame with a block not longer than its own order
    must be refused with SELA_HIP_ERANGE, exactly those."""
    from sela_amd import capi, codec

    o = oracle()
    rng = np.random.default_rng(77)
    landmarks = [1, 2, 3, 31, 63, 64, 65, 100, 101, 102, 103, 127, 128, 129, 2047, 2048, 2049]
    refused = coded = 0
    for trial in range(300):
        n = int(rng.choice(landmarks)) if trial % 3 == 0 else int(rng.integers(1, 6001))
        ch = int(rng.integers(1, 7))
        amp = int(2 ** rng.uniform(0, 20))
        t = np.arange(n)
        x = np.zeros((ch, n), np.int64)
        for c in range(ch):
            kind = rng.integers(0, 6)
            if kind == 0:
                x[c] = 0
            elif kind == 1:
                x[c] = rng.integers(-amp, amp + 1)
            elif kind == 2:
                x[c] = rng.integers(-amp, amp + 1, n)
            else:
                x[c] = np.round(amp * 0.7 * np.sin(t * rng.uniform(0.001, 1.5) + rng.uniform(0, 6)) + rng.normal(0, amp * rng.choice([0.0, 0.01, 0.2]), n))
        if ch == 2 and rng.random() < 0.5:
            x[1] = x[0] - rng.integers(-2, 3, n)
        x = np.clip(x, -(1 << 20), 1 << 20).astype(np.int32)
        # the orders of every block the frame encoder analyses (the difference signal of a stereo frame included)
        signals = [x[c] for c in range(ch)] + ([(x[0] - x[1]).astype(np.int32)] if ch == 2 else [])
        short = any(o.lpc_analyze(s)[0] >= n for s in signals)
        if short:
            with pytest.raises(capi.SelaHipError) as err:
                codec.encode_i32(x[None])
            assert err.value.code == -6, (trial, n, ch)
            refused += 1
            continue
        frames, offs = codec.encode_i32(x[None])
        want = o.frame_encode_i32(x)
        assert frames.tobytes() == want, (trial, n, ch, amp)
        dec = codec.decode_i32(frames, offs, ch)[0]
        ref_dec, used = o.frame_decode_i32(want, ch)
        assert used == len(want)
        for c in range(ch):
            assert np.array_equal(dec[c], ref_dec[c]), (trial, n, ch, c)
        coded += 1
    assert refused >= 5 and coded >= 200, (refused, coded)


def test_hostile_streams_through_the_any_length_decoder(gpu):  # noqa: F811
    """Frames no encoder wrote: random words as Rice streams, random parameters, orders and lengths, coefficient values far
    outside the dequantisation tables (clamped, like the oracle clamps them).  Where the oracle reads past a stream's end or
    a predictor coefficient leaves int64 the call must fail (EFORMAT / ERANGE); everywhere else every sample must be the
    oracle's -- wrap-around arithmetic, 32-bit results."""
    import ctypes as C
    import struct

    from sela_amd import capi, codec

    o = oracle()
    rng = np.random.default_rng(99)
    same = failed = 0
    for trial in range(250):
        ch = int(rng.integers(1, 4))
        subs = []
        n_frame = int(rng.integers(1, 700))
        roomy = rng.random() < 0.7  # (most frames get streams long enough for their values; the rest run dry)
        for c in range(ch):
            order = int(rng.integers(0, 101))
            ck = int(rng.integers(0, 12))
            rk = int(rng.integers(0, 20))
            n = n_frame if rng.random() < 0.8 else int(rng.integers(1, 700))
            cwords = (order * (ck + 3)) // 32 + 2 + int(rng.integers(0, 4)) if roomy else int(rng.integers(0, 40))
            rwords = (n * (rk + 3)) // 32 + 8 + int(rng.integers(0, 8)) if roomy else int(rng.integers(1, 1 + (n * (rk + 3)) // 32 + 8))
            dense = rng.random() < 0.5  # sparse words = short unary runs, so that most streams do hold their values
            mk = (lambda m: (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)
                             & (rng.integers(0, 1 << 32, m, dtype=np.uint64) if not dense else np.uint64(0xFFFFFFFF))).astype(np.uint32))
            cw, rw = mk(cwords), mk(rwords)
            typ = 1 if (c > 0 and rng.random() < 0.3) else 0
            parent = int(rng.integers(0, c)) if typ else c
            subs.append(struct.pack("<BBBBHB", c, typ, parent, ck, cwords, order) + cw.tobytes() + struct.pack("<BHH", rk, rwords, n) + rw.tobytes())
        blob = bytes.fromhex("00ff55aa") + b"".join(subs)
        fl = C.c_uint32(0)
        b = np.frombuffer(blob, np.uint8).copy()
        out = np.zeros((ch, 700), np.int32)
        counts = np.zeros(ch, np.uint32)
        used = o._fdec32(b, ch, out, 700, counts, C.byref(fl))
        assert used == len(blob)
        bad = fl.value & (8 | 2 | 32)  # RICE_OVERRUN, COEF_OVERFLOW, BAD_FRAME
        offs = np.array([0, len(blob)], np.uint64)
        if bad:
            with pytest.raises(capi.SelaHipError) as err:
                codec.decode_i32(b, offs, ch)
            assert err.value.code in (-5, -6), (trial, hex(fl.value))
            failed += 1
        else:
            dec = codec.decode_i32(b, offs, ch)[0]
            for c in range(ch):
                assert np.array_equal(dec[c], out[c, : int(counts[c])]), (trial, c, hex(fl.value))
            same += 1
    assert same >= 60 and failed >= 60, (same, failed)


def _subframe_words(blob, ch):
    """aligned words (coefficient words + 2 + residue words) of every subframe of one frame"""
    import struct

    out, p = [], 4
    for _ in range(ch):
        cw = struct.unpack_from("<H", blob, p + 4)[0]
        rw = struct.unpack_from("<H", blob, p + 7 + 4 * cw + 1)[0]
        out.append(cw + 2 + rw)
        p += 12 + 4 * (cw + rw)
    return out


def _both_decoders(blob, offs, ch):
    """decode_i32 with the standard kernel offered and with the any-length kernel alone -> (offered, alone, chunks the standard
    kernel decoded)."""
    from sela_amd import capi, codec

    lib = capi.lib()
    try:
        lib.sela_hip_debug_standard_first(1)
        before = lib.sela_hip_debug_standard_chunks()
        offered = codec.decode_i32(blob, offs, ch)
        took = lib.sela_hip_debug_standard_chunks() - before
        lib.sela_hip_debug_standard_first(0)
        alone = codec.decode_i32(blob, offs, ch)
    finally:
        lib.sela_hip_debug_standard_first(-1)
    return offered, alone, took


def test_standard_subframes_come_out_as_32_bit_samples_on_the_fast_parse_and_synthesis(gpu, kats):  # noqa: F811
    """sela_hip_decode_i32 (frame::FrameDecoder behind it) on 2048-sample frames: k_decode_subframes32 must give the oracle's 32-bit
    samples -- an encoder's frames (stereo with both decisions, mono, three channels; one call of many frames and calls of one),
    frames of samples far beyond 16 bits, and the KAT blocks -- and so must the any-length kernel on the same bytes."""
    from sela_amd import codec

    o = oracle()
    for pcm in (synth_frames(40, 2, 0), synth_frames(7, 1, 2), synth_frames(5, 3, 4)):
        ch = pcm.shape[2]
        frames, offs = codec.encode_host(pcm)
        offered, alone, took = _both_decoders(frames, offs, ch)
        assert took == 1
        for f in range(pcm.shape[0]):
            for c in range(ch):
                assert np.array_equal(offered[f][c], pcm[f, :, c].astype(np.int32)), (f, c)
                assert np.array_equal(alone[f][c], offered[f][c]), (f, c)
        one = codec.decode_i32(frames[int(offs[3]):int(offs[4])], _one(int(offs[4] - offs[3])), ch)[0]  # (the library's own choice)
        assert np.array_equal(np.stack(one, axis=1), pcm[3].astype(np.int32))
    # samples no WAV file holds: 21-bit tones and noise, a 32-bit frame API's business
    rng = np.random.default_rng(5)
    t = np.arange(2048)
    wide = np.stack([
        np.round((1 << 20) * 0.9 * np.sin(t * 0.01) + rng.normal(0, 3000, 2048)),
        np.round((1 << 19) * np.sin(t * 0.31 + 1) + rng.normal(0, 10, 2048)),
        rng.integers(-(1 << 20), 1 << 20, 2048).astype(np.float64),
    ]).astype(np.int32)
    taken = 0
    for chans in (wide[:1], wide[1:2], wide[:2], wide):
        frames, offs = codec.encode_i32(chans[None])
        want = o.frame_encode_i32(chans)
        assert frames.tobytes() == want
        ref_dec, used = o.frame_decode_i32(want, len(chans))
        offered, alone, took = _both_decoders(frames, offs, len(chans))
        # the fast kernel takes every clean frame: a subframe that fits the parser's plan (1072 aligned words) in one piece, the
        # uniform 21-bit noise of the third channel by segments
        assert took == 1 and used == len(want), (len(chans), _subframe_words(want, len(chans)))
        taken += took
        for c in range(len(chans)):
            assert np.array_equal(offered[0][c], ref_dec[c]) and np.array_equal(alone[0][c], ref_dec[c]), c
    assert taken >= 2
    names = [str(n) for n in kats["blk_names"]]
    blocks = np.stack([kats[f"blk/{n}/samples"] for n in names]).astype(np.int32)  # (diff_extreme is 17-bit)
    frames, offs = codec.encode_i32(blocks[:, None, :])
    taken = 0
    for i, name in enumerate(names):  # (a call per block: the standard kernel takes a chunk whole or not at all)
        blob = frames[int(offs[i]):int(offs[i + 1])]
        ref_dec, _ = o.frame_decode_i32(blob.tobytes(), 1)
        offered, alone, took = _both_decoders(blob, _one(len(blob)), 1)
        assert took == 1, name
        taken += took
        assert np.array_equal(offered[0][0], ref_dec[0]) and np.array_equal(alone[0][0], ref_dec[0]), name
    assert taken >= len(names) // 2


def test_chunks_that_mix_the_one_piece_parse_with_segments(gpu):  # noqa: F811
    """A chunk in which some subframes take the frame kernel's one-piece parse and others go by segments -- a frame of another
    length among 2048-sample ones, a Rice stream beyond the parser's plan (incompressible full-scale noise), a subframe type the
    reference ignores -- comes out of the fast kernel in one go: same answer as the serial kernel alone."""
    import struct

    from sela_amd import codec

    o = oracle()
    std = synth_frames(6, 2, 9)
    blobs = [o.frame_encode(std[f]) for f in range(6)]
    odd = synth_pcm(1000 * 2, 2, 3).reshape(2, 1000, 2)
    blobs.insert(3, o.frame_encode(odd[0]))
    stream = np.frombuffer(b"".join(blobs), np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.uint64)
    offered, alone, took = _both_decoders(stream, offs, 2)
    assert took == 1
    for f in range(len(blobs)):
        ref_dec, _ = o.frame_decode_i32(blobs[f], 2)
        for c in range(2):
            assert np.array_equal(offered[f][c], ref_dec[c]) and np.array_equal(alone[f][c], ref_dec[c]), (f, c)
    # beyond the plan: more than 1072 aligned words in one subframe
    noise = np.random.default_rng(2).integers(-(1 << 19), 1 << 19, (1, 2, 2048)).astype(np.int32)
    frames, fo = codec.encode_i32(noise)
    assert max(_subframe_words(frames.tobytes(), 2)) > 1072
    offered, alone, took = _both_decoders(frames, fo, 2)
    assert took == 1
    ref_dec, _ = o.frame_decode_i32(frames.tobytes(), 2)
    for c in range(2):
        assert np.array_equal(offered[0][c], ref_dec[c]) and np.array_equal(alone[0][c], ref_dec[c])
    # a subframe of a type the reference's two passes both skip (src/frame/frame_decoder.cpp:17-69): its channel stays empty
    blob = bytearray(blobs[0])
    second = 4 + 12 + 4 * (struct.unpack_from("<H", blob, 8)[0] + struct.unpack_from("<H", blob, 4 + 7 + 4 * struct.unpack_from("<H", blob, 8)[0] + 1)[0])
    assert blob[second] == 1  # (the second subframe's channel byte)
    blob[second + 1] = 7
    b = np.frombuffer(bytes(blob), np.uint8)
    ref_dec, used = o.frame_decode_i32(bytes(blob), 2)
    offered, alone, took = _both_decoders(b, _one(len(blob)), 2)
    assert took == 1 and used == len(blob)
    for c in range(2):
        assert np.array_equal(offered[0][c], ref_dec[c]) and np.array_equal(alone[0][c], ref_dec[c]), c


def test_hostile_2048_sample_frames_through_both_decoders(gpu):  # noqa: F811
    """Frames of 2048 samples no encoder wrote (random words as Rice streams, random parameters and orders, coefficient values
    outside the tables): whatever the standard kernel takes must be the oracle's wrap-around arithmetic in 32 bits, whatever it
    leaves alone the any-length kernel's answer; failures (a stream that runs dry, a coefficient beyond int64) fail both ways."""
    import ctypes as C
    import struct

    from sela_amd import capi, codec

    o = oracle()
    rng = np.random.default_rng(123)
    same = failed = taken = 0
    import os
    trials = int(os.environ.get("SELA_HOSTILE_TRIALS", "160"))  # (a long soak: SELA_HOSTILE_TRIALS=5000)
    for trial in range(trials):
        ch = int(rng.integers(1, 4))
        subs = []
        roomy = rng.random() < 0.75
        for c in range(ch):
            order = int(rng.integers(0, 101))
            ck = int(rng.integers(0, 10))
            rk = int(rng.integers(0, 14))
            n = 2048
            cwords = (order * (ck + 3)) // 32 + 2 + int(rng.integers(0, 4)) if roomy else int(rng.integers(0, 40))
            rwords = min(1040 - cwords, (n * (rk + 3)) // 32 + 8 + int(rng.integers(0, 8))) if roomy else int(rng.integers(1, 900))
            dense = rng.random() < 0.3
            mk = (lambda m: (rng.integers(0, 1 << 32, m, dtype=np.uint64) & rng.integers(0, 1 << 32, m, dtype=np.uint64)
                             & (rng.integers(0, 1 << 32, m, dtype=np.uint64) if not dense else np.uint64(0xFFFFFFFF))).astype(np.uint32))
            cw, rw = mk(cwords), mk(rwords)
            typ = 1 if (c > 0 and rng.random() < 0.3) else 0
            parent = int(rng.integers(0, c)) if typ else c
            subs.append(struct.pack("<BBBBHB", c, typ, parent, ck, cwords, order) + cw.tobytes() + struct.pack("<BHH", rk, rwords, n) + rw.tobytes())
        blob = bytes.fromhex("00ff55aa") + b"".join(subs)
        fl = C.c_uint32(0)
        b = np.frombuffer(blob, np.uint8).copy()
        out = np.zeros((ch, 2048), np.int32)
        counts = np.zeros(ch, np.uint32)
        used = o._fdec32(b, ch, out, 2048, counts, C.byref(fl))
        assert used == len(blob)
        bad = fl.value & (8 | 2 | 32)  # RICE_OVERRUN, COEF_OVERFLOW, BAD_FRAME
        offs = np.array([0, len(blob)], np.uint64)
        if bad:
            for mode in (1, 0):
                capi.lib().sela_hip_debug_standard_first(mode)
                try:
                    with pytest.raises(capi.SelaHipError) as err:
                        codec.decode_i32(b, offs, ch)
                finally:
                    capi.lib().sela_hip_debug_standard_first(-1)
                assert err.value.code in (-5, -6), (trial, mode, hex(fl.value))
            failed += 1
        else:
            offered, alone, took = _both_decoders(b, offs, ch)
            taken += took
            for c in range(ch):
                assert np.array_equal(offered[0][c], out[c, : int(counts[c])]), (trial, c, took, hex(fl.value))
                assert np.array_equal(alone[0][c], out[c, : int(counts[c])]), (trial, c, hex(fl.value))
            same += 1
    assert same >= 40 and failed >= 20 and taken >= 20, (same, failed, taken)
