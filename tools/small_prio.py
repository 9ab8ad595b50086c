import sys, time, torch
sys.path.insert(0, "/root/repo")
from sela_amd import capi, codec
from sela_amd.synth import synth_frames
lib = capi.lib()
lib.sela_hip_enable_kernel_timing(1)
for n in (250, 500, 1000, 2000, 2900):
    pcm = torch.from_numpy(synth_frames(n, 2, 1)).cuda()
    enc = codec.Encoder(n, 2)
    acc = {0: [], 0x00010203: []}
    for rep in range(6):
        for pr in acc:
            lib.sela_hip_debug_priorities(pr)
            for _ in range(6):
                enc.encode(pcm); torch.cuda.synchronize(); acc[pr].append(capi.kernel_times(3)[0])
    row = []
    for pr in acc:
        ks = sorted(acc[pr]); row.append(ks[len(ks)//2])
    print(n, "frames: k_encode_blocks %.4f ms without, %.4f ms with falling priorities" % tuple(row))
lib.sela_hip_debug_priorities(0)
