#!/usr/bin/env python3
"""tools/group_stamps.py -- where the time of ONE host-pointer encode goes, group of frames by group of frames.

The product kernels carry no instrumentation.  `build` (runs anywhere hipcc does) copies sela_amd/csrc to a scratch
directory, patches wall-clock stamps (s_memrealtime, 100 MHz) into the copy -- when the stager published a group's last
frame, when the group's last block accepted it, when the group's last arriver started, finished its look-back, had
issued its copy and had it acknowledged -- and builds a second libsela_hip.so under tools/tmp/stamped/.  `run` (on the
GPU box) runs host/sela_filebench against that library (LD_LIBRARY_PATH) and prints the table DESIGN.md 5.4 quotes.

    python tools/group_stamps.py build            # here
    gpurun -- 'python tools/group_stamps.py run'  # there; tools/tmp/ travels, it is not tracked

The patches are anchored on source lines and fail loudly when the source has moved on.
"""
import os
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "tmp", "stamped")
N_CELLS = 1024


def _n(text):
    return text.replace("N_CELLS", str(N_CELLS))


ENCODE_PATCHES = [(old, _n(new)) for old, new in [
    ("template <typename T>\n__device__ __forceinline__ void store_through(T* p, T v)",
     "__device__ unsigned long long g_stamps[6][N_CELLS];\ntemplate <typename T>\n__device__ __forceinline__ void store_through(T* p, T v)"),
    ("        if (lane == 0) // (nothing comes back out of this one-lane region)\n            store_through(pcm_ready + f, ((uint64_t)ticket << 32) | sum);",
     "        if (lane == 0) {\n            store_through(pcm_ready + f, ((uint64_t)ticket << 32) | sum);\n"
     "            atomicMax(&g_stamps[0][(f / kGroupFrames) % N_CELLS], wall_clock64());\n        }"),
    ("            if (wave_sum_small(sum) == (uint32_t)c) {\n                return true;",
     "            if (wave_sum_small(sum) == (uint32_t)c) {\n                if (lane == 0)\n"
     "                    atomicMax(&g_stamps[1][(f / kGroupFrames) % N_CELLS], wall_clock64());\n                return true;"),
    ("    __builtin_amdgcn_s_setprio(3);\n    const uint32_t f0 = g * kGroupFrames;",
     "    __builtin_amdgcn_s_setprio(3);\n    if (lane == 0)\n        g_stamps[2][g % N_CELLS] = wall_clock64();\n    const uint32_t f0 = g * kGroupFrames;"),
    ("    // ---- the bytes ----\n    if (nfg * fa.channels <= 64u) {",
     "    if (lane == 0)\n        g_stamps[3][g % N_CELLS] = wall_clock64();\n    // ---- the bytes ----\n    if (nfg * fa.channels <= 64u) {"),
    ("    if (fa.groups_done && lane == 0)\n        (void)group_arrive(fa.groups_done, fa.tag, n_groups);\n}",
     "    if (lane == 0)\n        g_stamps[4][g % N_CELLS] = wall_clock64();\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
     "    if (lane == 0)\n        g_stamps[5][g % N_CELLS] = wall_clock64();\n"
     "    if (fa.groups_done && lane == 0)\n        (void)group_arrive(fa.groups_done, fa.tag, n_groups);\n}"),
    ("#include <algorithm>\n", "#include <algorithm>\n#include <cstdio>\n#include <cstdlib>\n"),
]]
DUMP = _n('''
void dump_stamps()
{
    const char* path = std::getenv("SELA_STAMPS_FILE");
    if (!path)
        return;
    static unsigned long long h[6][N_CELLS], z[6][N_CELLS];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h) != hipSuccess)
        return;
    if (FILE* f = std::fopen(path, "wb")) {
        std::fwrite(h, 1, sizeof h, f);
        std::fclose(f);
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z); // (the stamps of a group's frames are maxima: start the next call from 0)
}
''')
CAPI_PATCHES = [
    ("namespace sela {", "namespace sela { void dump_stamps(); }\nnamespace sela {"),
    ("        return fail(SELA_HIP_EINVAL, \"bad argument\");\n    return job_end(job, frames_final, bytes_final);",
     "        return fail(SELA_HIP_EINVAL, \"bad argument\");\n    const int rc_stamped = job_end(job, frames_final, bytes_final);\n    sela::dump_stamps();\n    return rc_stamped;"),
]


def patch(path, patches, tail=None):
    s = open(path).read()
    for old, new in patches:
        assert s.count(old) >= 1, "anchor not found in %s:\n%s" % (os.path.basename(path), old)
        s = s.replace(old, new, 1)
    if tail:
        end = s.rindex("} // namespace sela")
        s = s[:end] + tail + s[end:]
    open(path, "w").write(s)


def build():
    src = tempfile.mkdtemp(prefix="stamped_csrc_")
    try:
        for f in os.listdir(os.path.join(ROOT, "sela_amd", "csrc")):
            shutil.copy(os.path.join(ROOT, "sela_amd", "csrc", f), src)
        patch(os.path.join(src, "sela_encode.hip"), ENCODE_PATCHES, DUMP)
        patch(os.path.join(src, "sela_capi.hip"), CAPI_PATCHES)
        os.makedirs(OUT, exist_ok=True)
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-I" + os.path.join(ROOT, "include")] + [os.path.join(src, f) for f in ("sela_encode.hip", "sela_decode.hip", "sela_capi.hip")] \
            + ["-o", os.path.join(OUT, "libsela_hip.so")]
        print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    finally:
        shutil.rmtree(src, ignore_errors=True)
    print("built", os.path.join(OUT, "libsela_hip.so"))


def report(path, n_frames):
    import numpy as np

    a = np.fromfile(path, dtype=np.uint64).reshape(6, N_CELLS).astype(np.int64)
    n = min((n_frames + 7) // 8, N_CELLS)
    pub, acc, start, looked, issued, acked = [a[i, :n] / 100.0 for i in range(6)]  # us
    t0 = pub.min()
    print("first group published at 0, last at %.1f us; last group acknowledged at %.1f us" % (pub.max() - t0, acked.max() - t0))
    for name, x, y in (("accept - publish", acc, pub), ("last arriver - accept (blocks)", start, acc), ("look-back", looked, start),
                       ("copy issued", issued, looked), ("copy acknowledged", acked, issued), ("group done - publish", acked, pub)):
        v = x - y
        print("%-32s mean %7.1f  median %7.1f  p90 %7.1f  max %7.1f   last 16 groups %7.1f" % (name, v.mean(), np.median(v), np.percentile(v, 90), v.max(), v[-16:].mean()))
    print("group:  published accepted last-arriver looked-back copy-issued acknowledged   (us from the first publication)")
    for g in list(range(0, min(4, n))) + list(range(max(n - 16, 4), n)):
        print("%5d " % g + " ".join("%9.1f" % (x[g] - t0) for x in (pub, acc, start, looked, issued, acked)))


def run():
    sys.path.insert(0, ROOT)
    import torch
    from sela_amd import synth

    n_frames = int(os.environ.get("STAMP_FRAMES", "3875"))
    exe = os.path.join(ROOT, "host", "sela_filebench")
    assert os.path.exists(os.path.join(OUT, "libsela_hip.so")), "run `python tools/group_stamps.py build` first"
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        pcm = synth.synth_frames_torch(n_frames, 2, 1, 0, torch.device("cuda")).cpu().numpy()
        data = pcm.astype("<i2").tobytes()
        wav = os.path.join(tmp, "track.wav")
        with open(wav, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IhHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16)
                    + b"data" + struct.pack("<I", len(data)) + data)
        stamps = os.path.join(tmp, "stamps.bin")
        env = dict(os.environ, LD_LIBRARY_PATH=OUT, SELA_STAMPS_FILE=stamps, SELA_FILEBENCH_ENCODE_ONLY="1")
        out = subprocess.run([exe, wav, tmp, "5", "e2e"], capture_output=True, text=True, timeout=120, env=env)
        print("stamped build:", out.stdout.strip(), out.stderr.strip()[-300:])
        report(stamps, n_frames)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    {"build": build, "run": run}.get(sys.argv[1] if len(sys.argv) > 1 else "", lambda: print(__doc__))()
