#!/usr/bin/env python3
"""How many losing stereo candidates leave their slot unwritten (sela_encode_tail.inc): the workspace is filled with a pattern,
one encode runs, the slots that still hold the pattern are counted.  python tools/skip_rate.py"""
import sys, numpy as np, torch
sys.path.insert(0,'.')
from sela_amd import capi, codec
from sela_amd.synth import synth_frames
lib=capi.lib()
for n,t in ((3875,-1),(1000,-1),(10000,-1),(20000,-1)):
    pcm=torch.from_numpy(synth_frames(n,2,0)).cuda()
    lib.sela_hip_debug_encode_teams(t)
    enc=codec.Encoder(n,2); enc.workspace.fill_(0xA5)
    enc.encode(pcm); torch.cuda.synchronize()
    ws=enc.workspace.cpu().numpy(); base=(-enc.workspace.data_ptr())%256
    slots_at=base+(n*3*8+255)//256*256
    slots=ws[slots_at:slots_at+n*3*2240*4].view(np.uint32).reshape(n*3,2240)
    un=(slots[:,32]==0xA5A5A5A5)&(slots[:,33]==0xA5A5A5A5)
    print(n, "kernel", lib.sela_hip_debug_encode_kernel(n,2), "losers skipped: %.1f %%"%(100*(un[1::3].sum()+un[2::3].sum())/n))
