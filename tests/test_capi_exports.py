"""CPU-only: the C-ABI library loads and exports exactly what include/sela_hip.h declares."""
import os
import re

import pytest

from sela_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="sela_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sela_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(capi.EXPORTS)
    assert _declared("sela_hip_debug.h") == sorted(capi.DEBUG_EXPORTS)
    assert not [n for n in _declared() if "debug" in n], "test hooks belong in sela_hip_debug.h, not in the boundary"


def test_library_exports_every_declared_symbol():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g

        g.build_hip_library()
    lib = capi.lib()
    for name in _declared() + _declared("sela_hip_debug.h"):
        assert hasattr(lib, name), name


def test_sizing_entry_points_need_no_gpu():
    lib = capi.lib()
    assert lib.sela_hip_signals_per_frame(1) == 1
    assert lib.sela_hip_signals_per_frame(2) == 3  # ch0, ch1, ch0-ch1 (src/frame/frame_encoder.cpp:18)
    assert lib.sela_hip_signals_per_frame(6) == 6
    assert lib.sela_hip_encode_workspace_bytes(10, 2) > 10 * 3 * 2240 * 4
    assert lib.sela_hip_encode_bound_bytes(1, 2) >= 4 + 2 * 12 + 2 * 4 * 2240


def test_index_frames_on_golden_stream(kats):
    import numpy as np

    from sela_amd import codec

    blobs = [kats["frame/stereo_same_sine/bytes"], kats["frame/stereo_synth_diff/bytes"], kats["frame/stereo_silence/bytes"]]
    stream = np.concatenate(blobs)
    offs = codec.index_frames(stream, 3, 2)
    assert offs.tolist() == [0, len(blobs[0]), len(blobs[0]) + len(blobs[1]), len(stream)]
    # a corrupted sync word silently ends the walk (src/file/sela_file.cpp:54-56)
    bad = stream.copy()
    bad[len(blobs[0])] ^= 0xFF
    assert codec.index_frames(bad, 3, 2).tolist() == [0, len(blobs[0])]


def test_no_gpu_means_loud_failure():
    import numpy as np

    lib = capi.lib()
    if lib.sela_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    from sela_amd import codec

    with pytest.raises(capi.SelaHipError) as e:
        codec.encode_host(np.zeros((1, 2048, 2), np.int16))
    assert e.value.code == -1  # ENODEV: no CPU fallback


def test_index_samples_and_the_any_length_bound_need_no_gpu(generic_kats, kats):
    """sela_hip_index_samples walks a stream's headers on the host: frames of different lengths, a frame whose channels
    disagree (the first subframe's count stands for the frame), a stream the walk falls off."""
    import numpy as np

    from sela_amd import codec

    lib = capi.lib()
    a = generic_kats["n128_stereo_diff_i16/bytes"]
    b = generic_kats["n1000_stereo_indep_i17/bytes"]
    c = kats["frame/stereo_same_sine/bytes"]
    stream = np.concatenate([a, b, c, a])
    offs = np.cumsum([0, len(a), len(b), len(c), len(a)]).astype(np.uint64)
    so, largest = codec.index_samples(stream, offs, 2)
    assert largest == 2048 and so.tolist() == [0, 128, 1128, 3176, 3304]
    mixed = generic_kats["crafted/mixed_lengths/bytes"]
    so, largest = codec.index_samples(mixed, np.array([0, len(mixed)], np.uint64), 3)
    assert largest == 300 and so.tolist() == [0, 300]
    so, largest = codec.index_samples(a, np.array([0, len(a) - 7], np.uint64), 2)
    assert largest == 0  # (a frame cut short: the walk runs off its end; the decode calls report EFORMAT)
    assert lib.sela_hip_encode_bound_bytes_n(3, 2, 2048) == lib.sela_hip_encode_bound_bytes(3, 2)
    for n, ch in ((1, 1), (128, 2), (1000, 3), (65535, 2)):
        bound = lib.sela_hip_encode_bound_bytes_n(1, ch, n)
        assert bound >= 4 + ch * (12 + 4 * min(65535, n)) and bound <= 4 + ch * (12 + 4 * (32 + 65535))


def test_no_gpu_means_loud_failure_on_the_any_length_route_too():
    import numpy as np

    lib = capi.lib()
    if lib.sela_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    from sela_amd import codec

    for call in (lambda: codec.encode_host(np.zeros((1, 1000, 2), np.int16)), lambda: codec.encode_i32(np.zeros((1, 2, 300), np.int32)),
                 lambda: codec.lpc_encode_n(np.zeros((1, 500), np.int32)), lambda: codec.lpc_decode_n(np.ones(1, np.int32), np.zeros((1, 100), np.int32), np.zeros((1, 500), np.int32))):
        with pytest.raises(capi.SelaHipError) as e:
            call()
        assert e.value.code == -1 and "no CPU fallback" in str(e.value)
