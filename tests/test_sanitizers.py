"""Sanitizer builds (no GPU): the C restatement under ASan + UBSan on the known-answer tests, the C++ host's container /
I/O-pool self-test under ASan + UBSan and under TSan, and the call coalescer of libsela_hip.so (sela_amd/csrc/
sela_coalescer.h) on a CPU stub backend under TSan.  What SURVEY section 5 planned where the reference has only a coverage
build (/root/reference/CMakeLists.txt:146-158)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    assert r.returncode == 0, f"{' '.join(map(str, cmd))}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r


def _host_lib_or_skip():
    if not os.path.exists(os.path.join(ROOT, "sela_amd", "libsela_hip.so")):
        pytest.skip("libsela_hip.so is not built")


def test_oracle_kats_under_asan_and_ubsan():
    """oracle/sela_oracle.c compiled with -fsanitize=address,undefined (no recovery) runs the golden / KAT tests clean."""
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "_asan", "libsela_oracle.so")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, SELA_ORACLE_LIB=lib, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_selftest_under_sanitizers(tmp_path, kind):
    """host/ (frame / file / codec / player classes, the I/O pool with its read-ahead and write-behind strands, the parallel
    object builders) built with -fsanitize=address,undefined and with -fsanitize=thread: the container sections of
    host_selftest -- everything that runs without a GPU -- finish clean."""
    _host_lib_or_skip()
    _run(["make", "-C", HOST, kind])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(HOST, f"build_{kind}", "host_selftest"), str(tmp_path)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "selftest: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_call_coalescer_under_tsan(tmp_path):
    """The group-commit coalescer behind sela_hip_encode / sela_hip_decode, instantiated on a CPU stub instead of the device
    (the seam is a template parameter: nothing of the stub is in libsela_hip.so): 16 threads, small calls of two channel
    counts for two devices, too-small buffers and malformed frames among them -- every call gets its own result, no race;
    with the shipped number of batches in flight (two), with one and with three, and the per-call wake-ups of round 6."""
    exe = tmp_path / "coalescer_tsan"
    _run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "include"),
          "-I" + os.path.join(ROOT, "sela_amd", "csrc"), os.path.join(ROOT, "tests", "c", "coalescer_stress.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe), "16", "120"], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and r.stdout.count(" 0 failures") == 3 and "ThreadSanitizer" not in r.stderr, (r.stdout, r.stderr[-3000:])
    # the seam stays out of the product: the library knows nothing of the stub
    lib = os.path.join(ROOT, "sela_amd", "libsela_hip.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-DC", lib], capture_output=True, text=True).stdout
        assert "StubBackend" not in syms
