"""ur_gym_amd — MI355X-native vectorised UR5e reach environments (drop-in for the step()/reset() hot path of
WanqingXia/UR-gym: UR5OriReach-v1, UR5ObsReach-v1, UR5DynReach-v1).

The compute path is hand-written HIP for gfx950 behind the C-ABI of include/urgym.h; this package is the thin
Python host side (PyTorch-ROCm tensors for device memory and streams).  Importing the package does not load the
native library; constructing an environment does, and raises if it is missing (no CPU fallback).
"""
from ._abi import ENV_IDS  # noqa: F401

__all__ = ["ENV_IDS", "UR5ReachVectorEnv", "make_vec"]


def __getattr__(name):  # lazy: torch is only imported when an environment is requested
    if name in ("UR5ReachVectorEnv", "make_vec"):
        from . import vector_env

        return getattr(vector_env, name)
    raise AttributeError(name)
