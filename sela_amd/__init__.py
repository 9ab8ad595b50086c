"""sela_amd -- MI355X-native SELA frame encode/decode path.

The product is the C-ABI library `libsela_hip.so` (include/sela_hip.h, sources in
sela_amd/csrc/).  This package is the thin Python plumbing around it used by the tests, the
benchmark and the multi-GPU sharding helper: ctypes bindings (`capi`), torch-tensor wrappers
(`codec`), the integer-only synthetic PCM generator (`synth`) and frame sharding over
torch.distributed (`sharding`).  Nothing here computes the codec on the CPU.
"""

__all__ = ["capi", "codec", "synth", "sharding"]
