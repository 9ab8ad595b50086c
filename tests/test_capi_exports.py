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
