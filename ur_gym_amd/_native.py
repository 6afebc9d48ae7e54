"""Loader of the HIP extension (liburgym_hip.so) through ctypes.

There is deliberately NO fallback: if the shared library is missing, cannot be loaded, or does not export the
ABI of include/urgym.h, importing the native layer raises.  The CPU oracle under oracle/ is test infrastructure
and is never reachable from here.
"""
import ctypes as C
import os
import subprocess

from . import _abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# URGYM_LIB: another build of the same extension (tuning experiments: tools/exp_*.sh); there is still no fallback of any kind
LIB_PATH = os.environ.get("URGYM_LIB") or os.path.join(_CSRC, "liburgym_hip.so")
_lib = None


class NativeError(RuntimeError):
    pass


def build(force=False):
    """Compile the extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-s"] + (["-B"] if force else [])
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  Import torch
    # FIRST so that this extension binds to the very runtime instance whose streams and device pointers it is handed;
    # loading the extension first would pull in a second, system-wide runtime.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ur_gym_amd/csrc`). There is no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime, wrong arch, ...
        raise NativeError(f"cannot load {LIB_PATH}: {e}. There is no CPU fallback.") from e
    missing = [s for s in _abi.EXPORTED_SYMBOLS if not hasattr(L, s)]
    if missing:
        raise NativeError(f"{LIB_PATH} does not export {missing}")
    L.urgym_abi_version.restype = C.c_int
    if L.urgym_abi_version() != _abi.ABI_VERSION:
        raise NativeError(f"ABI mismatch: library {L.urgym_abi_version()} vs binding {_abi.ABI_VERSION}")
    L.urgym_config_default.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.Config)]
    L.urgym_obs_dims.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.urgym_create.argtypes = [C.POINTER(_abi.Config), C.c_int, C.POINTER(C.c_void_p)]
    L.urgym_destroy.argtypes = [C.c_void_p]
    L.urgym_bind.argtypes = [C.c_void_p, C.POINTER(_abi.Buffers)]
    L.urgym_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.urgym_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.urgym_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.urgym_refresh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.urgym_invalidate_records.argtypes = [C.c_void_p]
    L.urgym_derive_obstacle_motion.argtypes = [C.c_void_p, C.c_void_p]
    L.urgym_probe_closest.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.urgym_probe_pose_distance.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.urgym_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.urgym_query_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.urgym_query_refill_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.urgym_last_error.argtypes = [C.c_void_p]
    L.urgym_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(rc, handle=None):
    if rc != 0:
        msg = lib().urgym_last_error(handle)
        raise NativeError(f"urgym call failed ({rc}): {msg.decode() if msg else '?'}")
