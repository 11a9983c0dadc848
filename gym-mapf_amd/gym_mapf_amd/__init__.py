"""gym_mapf_amd -- MI355X-native drop-in for gym-mapf's batched ``MapfEnv.step()`` path.

Module layout mirrors the reference package so ``gym_mapf.envs.X`` becomes
``gym_mapf_amd.envs.X``; the hot path runs in ``lib/libmapf_hip.so`` (hand-written HIP
kernels for gfx950 behind the C ABI of ``include/mapf_hip.h``).  There is no CPU fallback.
"""
name = "gym_mapf_amd"
__version__ = "0.1.0"
