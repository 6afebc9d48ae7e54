"""ctypes binding of oracle/liburgym_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (ur_gym_amd) never does.  The oracle works on host numpy arrays laid out exactly like the device
buffers of include/urgym.h, so a test can hand the same state to both sides.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ur_gym_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liburgym_oracle.so")
_lib = None

_NP = {C.c_double: np.float64, C.c_float: np.float32, C.c_int32: np.int32, C.c_uint8: np.uint8}


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "urgym_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, fp, u8p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        L.urgym_oracle_config_default.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.Config)]
        L.urgym_oracle_create.argtypes = [C.POINTER(_abi.Config), C.POINTER(C.c_void_p)]
        L.urgym_oracle_destroy.argtypes = [C.c_void_p]
        L.urgym_oracle_bind.argtypes = [C.c_void_p, C.POINTER(_abi.Buffers)]
        L.urgym_oracle_reset.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int]
        L.urgym_oracle_refresh.argtypes = [C.c_void_p, u8p, C.c_int]
        L.urgym_oracle_step.argtypes = [C.c_void_p, fp, C.c_int]
        L.urgym_oracle_fk.argtypes = [dp, dp]
        L.urgym_oracle_ee_pose.argtypes = [dp, dp]
        L.urgym_oracle_distance.argtypes = [dp, dp]
        L.urgym_oracle_distance.restype = C.c_double
        L.urgym_oracle_angular_distance.argtypes = [dp, dp]
        L.urgym_oracle_angular_distance.restype = C.c_double
        L.urgym_oracle_quat_from_euler.argtypes = [dp, dp]
        L.urgym_oracle_euler_from_quat.argtypes = [dp, dp]
        L.urgym_oracle_dyn_velocity.argtypes = [dp, dp, C.c_double, dp]
        L.urgym_oracle_closest.argtypes = [C.c_int, dp, dp, C.c_int, dp, dp, C.c_double, dp]
        L.urgym_oracle_query.argtypes = [dp, dp, C.c_int, C.c_double, C.c_int, C.c_int, dp, C.POINTER(C.c_int)]
        L.urgym_oracle_integrate_obstacle.argtypes = [dp, dp, C.c_double]
        L.urgym_oracle_set_primitive_margin.argtypes = [C.c_double]
        L.urgym_oracle_last_epa_iterations.restype = C.c_int
        L.urgym_oracle_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, dp]
        L.urgym_oracle_hardware_threads.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def default_config(env_kind, num_envs):
    cfg = _abi.Config()
    rc = lib().urgym_oracle_config_default(env_kind, num_envs, C.byref(cfg))
    assert rc == 0
    return cfg


def alloc_buffers(env_kind, num_envs):
    """Zero-initialised host arrays with the shapes of include/urgym.h."""
    od, gd = _abi.OBS_DIMS[env_kind]
    return {name: np.zeros(shape(num_envs, od, gd), dtype=_NP[ct]) for name, ct, shape in _abi.BUFFER_FIELDS}


class OracleEnv:
    """The oracle behind the same reset/step/refresh verbs as the HIP library."""

    def __init__(self, env_kind, num_envs, threads=1, **overrides):
        self.env_kind, self.num_envs, self.threads = env_kind, num_envs, threads
        self.cfg = default_config(env_kind, num_envs)
        for k, v in overrides.items():
            setattr(self.cfg, k, v)
        self.obs_dim, self.goal_dim = _abi.OBS_DIMS[env_kind]
        self.buf = alloc_buffers(env_kind, num_envs)
        self._h = C.c_void_p()
        assert lib().urgym_oracle_create(C.byref(self.cfg), C.byref(self._h)) == 0
        self._bind()

    def _bind(self):
        b = _abi.Buffers()
        for name, ct, _ in _abi.BUFFER_FIELDS:
            arr = self.buf[name]
            assert arr.flags["C_CONTIGUOUS"]
            setattr(b, name, arr.ctypes.data_as(C.POINTER(ct)))
        self._cbuf = b
        assert lib().urgym_oracle_bind(self._h, C.byref(b)) == 0

    def load_state(self, state):
        """Copy arrays (same names as the buffers) into the oracle's buffers."""
        derive = False
        for k, v in state.items():
            v = np.asarray(v)
            if k == "obst_vel" and v.shape[0] == 6:  # a twist only: rows 6..8 (displacement per env step) are derived from it
                self.buf[k][:6] = v.reshape(6, self.num_envs)
                derive = True
            else:
                self.buf[k][...] = v.reshape(self.buf[k].shape)
        if derive:
            assert lib().urgym_oracle_derive_obstacle_motion(self._h) == 0

    def reset(self, seed=_abi.KEEP_SEED, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        mp = None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8))
        assert lib().urgym_oracle_reset(self._h, mp, C.c_uint64(seed), self.threads) == 0

    def refresh(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        mp = None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8))
        assert lib().urgym_oracle_refresh(self._h, mp, self.threads) == 0

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.num_envs, 6)
        assert lib().urgym_oracle_step(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), self.threads) == 0

    def close(self):
        if self._h:
            lib().urgym_oracle_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- scalar probes -----------------------------------------------------------------------------------------
def fk(q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    out = np.zeros((7, 12))
    lib().urgym_oracle_fk(_dp(q), _dp(out))
    return out[:, :9].reshape(7, 3, 3), out[:, 9:]


def ee_pose(q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    out = np.zeros(6)
    lib().urgym_oracle_ee_pose(_dp(q), _dp(out))
    return out


def distance(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().urgym_oracle_distance(_dp(a), _dp(b))


def angular_distance(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().urgym_oracle_angular_distance(_dp(a), _dp(b))


def quat_from_euler(rpy):
    rpy = np.ascontiguousarray(rpy, dtype=np.float64)
    out = np.zeros(4)
    lib().urgym_oracle_quat_from_euler(_dp(rpy), _dp(out))
    return out


def euler_from_quat(q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    out = np.zeros(3)
    lib().urgym_oracle_euler_from_quat(_dp(q), _dp(out))
    return out


def dyn_velocity(start, end, T=2.0):
    s = np.ascontiguousarray(start, dtype=np.float64)
    e = np.ascontiguousarray(end, dtype=np.float64)
    out = np.zeros(6)
    lib().urgym_oracle_dyn_velocity(_dp(s), _dp(e), T, _dp(out))
    return out


HULL, CYLZ, BOX, SPHERE = 0, 1, 2, 3


def closest(type_a, par_a, pose_a, type_b, par_b, pose_b, threshold=5.0):
    """pose = xyz + quaternion xyzw.  Returns dict(has_point, distance, penetrating, iterations)."""
    pa = np.zeros(3)
    pa[:len(par_a)] = par_a
    pb = np.zeros(3)
    pb[:len(par_b)] = par_b
    xa = np.ascontiguousarray(pose_a, dtype=np.float64)
    xb = np.ascontiguousarray(pose_b, dtype=np.float64)
    out = np.zeros(4)
    lib().urgym_oracle_closest(type_a, _dp(pa), _dp(xa), type_b, _dp(pb), _dp(xb), threshold, _dp(out))
    return dict(has_point=bool(out[0]), distance=out[1], penetrating=bool(out[2]), iterations=int(out[3]))


def query(q, obst_pose=None, margin=0.01, gjk_start=0, scope=0):
    q = np.ascontiguousarray(q, dtype=np.float64)
    ld = np.zeros(5)
    coll = C.c_int(0)
    if obst_pose is None:
        op = np.zeros(7)
        has = 0
    else:
        op = np.ascontiguousarray(obst_pose, dtype=np.float64)
        has = 1
    status = lib().urgym_oracle_query(_dp(q), _dp(op), has, margin, int(gjk_start), int(scope), _dp(ld), C.byref(coll))
    return ld, bool(coll.value), status


def integrate_obstacle(pos_quat, vel6, dt=0.04):
    """One env step (20 Bullet sub-steps) of the obstacle base; pos_quat = xyz + quaternion xyzw."""
    pq = np.ascontiguousarray(pos_quat, dtype=np.float64).copy()
    v = np.ascontiguousarray(vel6, dtype=np.float64)
    lib().urgym_oracle_integrate_obstacle(_dp(pq), _dp(v), dt)
    return pq


def set_primitive_margin(m):
    """What-if (tests only): collision margin of cylinders / boxes; < 0 restores Bullet's 0.001."""
    lib().urgym_oracle_set_primitive_margin(float(m))


def set_collision_groups(bits):
    """What-if (ablation tool only): bit 0 obstacle, bit 1 table + track, bit 2 self pairs; 7 = the reference."""
    lib().urgym_oracle_set_collision_groups(int(bits))


def last_epa_iterations():
    return lib().urgym_oracle_last_epa_iterations()


def philox(seed, env, episode, attempt):
    out = np.zeros(20)
    lib().urgym_oracle_philox(C.c_uint64(seed), env, episode, attempt, _dp(out))
    return out


def hardware_threads():
    return lib().urgym_oracle_hardware_threads()
