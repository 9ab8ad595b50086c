import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(autouse=True)
def _child_processes_are_bounded(monkeypatch):
    """Every child a test starts with subprocess.run (the CLI, the bound reference binaries, torch.distributed.run) gets a
    limit of five minutes unless the test sets its own: a child that hangs fails ITS test (TimeoutExpired, the child is
    killed) and the run goes on."""
    import subprocess

    real_run = subprocess.run

    def run(*args, **kwargs):
        kwargs.setdefault("timeout", 300)
        return real_run(*args, **kwargs)

    monkeypatch.setattr(subprocess, "run", run)


@pytest.fixture(scope="session")
def kats():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "kats.npz"))


@pytest.fixture(scope="session")
def digests():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
        return json.load(f)


def _load_json(name):
    import json

    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def file_digests():
    """Whole-file SHA-256s written by the reference's own sela::Encoder/Decoder + file classes."""
    return _load_json("file_digests.json")


@pytest.fixture(scope="session")
def album_digests():
    """BASELINE.json configs[3]: per-track .sela file / decoded PCM SHA-256s from the reference."""
    return _load_json("album_digests.json")


@pytest.fixture(scope="session")
def generic_digests():
    """Frames of any length / 17-bit samples through the reference's frame classes (make_golden.py generic)."""
    return _load_json("generic.json")


@pytest.fixture(scope="session")
def generic_kats():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "generic_kats.npz"))


@pytest.fixture(scope="session")
def ragged_digests():
    """Frames whose channels differ in length through the reference's frame::FrameEncoder (make_golden.py ragged)."""
    return _load_json("ragged.json")


@pytest.fixture(scope="session")
def ragged_kats():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "ragged_kats.npz"))
