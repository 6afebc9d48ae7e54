"""Vectorised UR5e reach environments on one MI355X: the Gymnasium-VectorEnv-shaped surface of the HIP path.

Mirrors, batched over ``num_envs`` environments, the Env API of the reference's ``RobotTaskEnv``
(UR_gym/envs/core.py:222-320) as ``train.py:39-60`` / ``demo.py:6-17`` / ``model_test.py:27-49`` consume it:

    reset(seed=None, options=None) -> (obs_dict, info)                      core.py:263-273
    step(actions[N,6]) -> (obs_dict, reward[N], terminated[N], truncated[N], info)   core.py:303-317 + TimeLimit(100)

Host Python only holds the per-env state as PyTorch-ROCm tensors and launches the fused HIP kernels through the
C-ABI (include/urgym.h); every number is computed on the GPU.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi, _native

try:  # gymnasium is optional (it is not installed in the build image)
    from gymnasium import spaces as _spaces

    Box, DictSpace = _spaces.Box, _spaces.Dict
except Exception:  # pragma: no cover - exercised when gymnasium is absent

    class Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low = np.full(shape, low, dtype=dtype)
            self.high = np.full(shape, high, dtype=dtype)
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class DictSpace(dict):
        @property
        def spaces(self):
            return self

        def sample(self):
            return {k: v.sample() for k, v in self.items()}


_TORCH_DTYPE = {C.c_double: torch.float64, C.c_float: torch.float32, C.c_int32: torch.int32, C.c_uint8: torch.uint8}


class UR5ReachVectorEnv:
    """N independent UR5{Ori,Obs,Dyn}Reach-v1 environments stepped by one fused kernel launch.

    Observations are returned as views of persistent device tensors (zero-copy); they are overwritten by the next
    ``step``/``reset``.  Pass ``copy_obs=True`` to get fresh tensors instead.
    """

    metadata = {"render_modes": []}

    def __init__(self, env_id="UR5DynReach-v1", num_envs=1, device="cuda:0", seed=0, auto_reset=True,
                 check_collision=True, copy_obs=False, **config_overrides):
        if env_id not in _abi.ENV_IDS:
            raise ValueError(f"unknown env id {env_id!r}; available: {sorted(_abi.ENV_IDS)}")
        self.lib = _native.lib()  # raises if the HIP extension is missing
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _native.NativeError("UR5ReachVectorEnv needs a ROCm GPU (torch device 'cuda:N'); there is no CPU path")
        self.env_id, self.env_kind = env_id, _abi.ENV_IDS[env_id]
        self.num_envs = int(num_envs)
        self.copy_obs = copy_obs
        self.obs_dim, self.goal_dim = _abi.OBS_DIMS[self.env_kind]
        self.cfg = _abi.Config()
        _native.check(self.lib.urgym_config_default(self.env_kind, self.num_envs, C.byref(self.cfg)))
        self.cfg.auto_reset = int(bool(auto_reset))
        self.cfg.check_collision = int(bool(check_collision))
        for k, v in config_overrides.items():
            if not hasattr(self.cfg, k):
                raise TypeError(f"unknown config field {k!r}")
            setattr(self.cfg, k, v)
        self._seed = int(seed)
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        _native.check(self.lib.urgym_create(C.byref(self.cfg), dev_index, C.byref(self._h)))
        # all buffers are torch tensors owned here; the C side only borrows the pointers
        self.buf = {}
        cb = _abi.Buffers()
        for name, ct, shape in _abi.BUFFER_FIELDS:
            t = torch.zeros(shape(self.num_envs, self.obs_dim, self.goal_dim), dtype=_TORCH_DTYPE[ct], device=self.device)
            self.buf[name] = t
            setattr(cb, name, C.cast(t.data_ptr(), C.POINTER(ct)))
        self._cbuf = cb
        _native.check(self.lib.urgym_bind(self._h, C.byref(cb)), self._h)
        # spaces (core.py:241-248, UR5.py:251)
        self.single_observation_space = DictSpace(
            observation=Box(-10.0, 10.0, (self.obs_dim,), np.float32),
            achieved_goal=Box(-10.0, 10.0, (self.goal_dim,), np.float32),
            desired_goal=Box(-10.0, 10.0, (self.goal_dim,), np.float32),
        )
        self.single_action_space = Box(-1.0, 1.0, (6,), np.float32)
        self.observation_space = DictSpace(
            observation=Box(-10.0, 10.0, (self.num_envs, self.obs_dim), np.float32),
            achieved_goal=Box(-10.0, 10.0, (self.num_envs, self.goal_dim), np.float32),
            desired_goal=Box(-10.0, 10.0, (self.num_envs, self.goal_dim), np.float32),
        )
        self.action_space = Box(-1.0, 1.0, (self.num_envs, 6), np.float32)
        self._needs_reset = True

    # ------------------------------------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _obs(self):
        keys = ("observation", "achieved_goal", "desired_goal")
        if self.copy_obs:
            return {k: self.buf[k].clone() for k in keys}
        return {k: self.buf[k] for k in keys}

    def _mask_ptr(self, ids_or_mask):
        if ids_or_mask is None:
            return None, None
        m = torch.as_tensor(ids_or_mask, device=self.device)
        if m.dtype == torch.bool or (m.dtype == torch.uint8 and m.numel() == self.num_envs):
            mask = m.to(torch.uint8).contiguous()
        else:
            mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
            mask[m.long()] = 1
        return mask, C.c_void_p(mask.data_ptr())

    # ------------------------------------------------------------------------------------------------ Env API
    def reset(self, seed=None, options=None, mask=None):
        """core.py:263-273 for every env (or the masked subset). ``seed`` re-keys the counter-based sampler."""
        if seed is not None:
            self._seed = int(seed)
            s = C.c_uint64(self._seed)
            if mask is None:
                # Gymnasium seeding contract: reset(seed=s) starts the SAME episodes every time.  The sampler is keyed by
                # (seed, env, episode id), so a seeded full reset rewinds the episode counters.
                self.buf["episode_id"].zero_()
        elif self._needs_reset:  # first reset, or the first one after SB3VecEnvAdapter.seed(): same rule
            s = C.c_uint64(self._seed)
            if mask is None:
                self.buf["episode_id"].zero_()
        else:
            s = C.c_uint64(_abi.KEEP_SEED)
        keep, mp = self._mask_ptr(mask)
        _native.check(self.lib.urgym_reset(self._h, mp, s, self._stream()), self._h)
        self._needs_reset = False
        info = {"is_success": self.buf["is_success"].bool()}
        return self._obs(), info

    def step(self, actions):
        """core.py:303-317 + TimeLimit; finished envs are auto-reset (their terminal observation is in
        info['final_observation'], valid where info['_final_observation'])."""
        if self._needs_reset:
            raise RuntimeError("call reset() before step()")
        a = torch.as_tensor(actions, device=self.device)
        if a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(torch.float32).contiguous()
        if a.shape != (self.num_envs, 6):
            raise ValueError(f"actions must have shape ({self.num_envs}, 6), got {tuple(a.shape)}")
        _native.check(self.lib.urgym_step(self._h, C.c_void_p(a.data_ptr()), self._stream()), self._h)
        b = self.buf
        # the flag buffers hold 0 / 1 bytes: reinterpret them as bool instead of launching a conversion kernel per flag
        terminated, truncated = b["terminated"].view(torch.bool), b["truncated"].view(torch.bool)
        # status: the per-env URGYM_STATUS_* word (sticky): penetration depths consumed, joint limits passed, NaN, ... (include/urgym.h)
        info = {"is_success": b["is_success"].view(torch.bool), "collision": b["collision"].view(torch.bool), "status": b["status"]}
        if self.copy_obs:  # like the observations: private copies on request, zero-copy views of the live buffers otherwise
            terminated, truncated = terminated.clone(), truncated.clone()
            info = {k: v.clone() for k, v in info.items()}
        if self.cfg.auto_reset:
            if self.copy_obs:  # private copies: the mask is materialised now, it must not read the flags of a later step
                info["_final_observation"] = terminated | truncated
                info["final_observation"] = {"observation": b["final_observation"].clone(), "achieved_goal": b["final_achieved_goal"].clone(),
                                             "desired_goal": b["final_desired_goal"].clone()}
            else:
                # zero-copy mode: like the observations, these are views of live buffers, valid until the next step() / reset();
                # the mask is derived on first access (a kernel launch only if somebody looks) -- read it before stepping again
                info = _LazyInfo(info, {"_final_observation": lambda: terminated | truncated})
                info["final_observation"] = {"observation": b["final_observation"], "achieved_goal": b["final_achieved_goal"],
                                             "desired_goal": b["final_desired_goal"]}
        reward = b["reward"].clone() if self.copy_obs else b["reward"]
        return self._obs(), reward, terminated, truncated, info

    def rollout(self, actions):
        """K fused steps without returning to Python in between; ``actions`` is [K, N, 6] float32 on the device."""
        a = torch.as_tensor(actions, device=self.device, dtype=torch.float32).contiguous()
        if a.dim() != 3 or a.shape[1:] != (self.num_envs, 6):
            raise ValueError("actions must be [K, N, 6]")
        _native.check(self.lib.urgym_rollout(self._h, C.c_void_p(a.data_ptr()), int(a.shape[0]), self._stream()), self._h)

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self.lib.urgym_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --------------------------------------------------------------------------------- task-level setters (a13)
    def set_goal(self, ids, goal):
        """ReachOri.set_goal (reach.py:202-204) for env ids; goal is [len(ids), 6] (xyz + rpy)."""
        ids = torch.as_tensor(ids, device=self.device).long()
        g = torch.as_tensor(goal, device=self.device, dtype=torch.float64).reshape(len(ids), -1)
        self.buf["goal"][: g.shape[1], ids] = g.T
        self._refresh(ids)

    def set_goal_and_obstacle(self, ids, data):
        """Reach{Obs,Dyn}.set_goal_and_obstacle (reach.py:328-335, 702-713).
        Obs: data = [goal xyz, obstacle xyz+rpy] (9); Dyn: [goal 6, obstacle_start 6, obstacle_end 6] (18);
        Sta: [goal 6, obstacle 6] (12) or the 18-column moving form."""
        ids = torch.as_tensor(ids, device=self.device).long()
        d = torch.as_tensor(data, device=self.device, dtype=torch.float64).reshape(len(ids), -1)
        if self.env_kind == _abi.ENV_OBS:
            assert d.shape[1] == 9
            self.buf["goal"][:3, ids] = d[:, :3].T
            self.buf["obst_start"][:, ids] = d[:, 3:9].T
        elif self.env_kind == _abi.ENV_STA:  # reach.py:484-507: 12 columns = static obstacle, 18 = start/end (moving)
            assert d.shape[1] in (12, 18)
            self.buf["goal"][:, ids] = d[:, :6].T
            self.buf["obst_start"][:, ids] = d[:, 6:12].T
            self.buf["obst_end"][:, ids] = d[:, 12:18].T if d.shape[1] == 18 else 0.0
        elif self.env_kind == _abi.ENV_DYN:
            assert d.shape[1] == 18
            self.buf["goal"][:, ids] = d[:, :6].T
            self.buf["obst_start"][:, ids] = d[:, 6:12].T
            self.buf["obst_end"][:, ids] = d[:, 12:18].T
        else:
            raise TypeError("UR5OriReach-v1 has no obstacle; use set_goal")
        self._refresh(ids)

    def _refresh(self, ids=None):
        keep, mp = self._mask_ptr(ids)
        _native.check(self.lib.urgym_refresh(self._h, mp, self._stream()), self._h)

    STATE_KEYS = ("q", "goal", "obst_start", "obst_end", "obst_pos", "obst_quat", "obst_vel", "link_dist", "step_count",
                  "episode_id")

    def get_state(self):
        """Snapshot of the per-env state tensors (SoA layout of include/urgym.h), on the host."""
        return {k: self.buf[k].detach().cpu().numpy().copy() for k in self.STATE_KEYS}

    def set_state(self, state, refresh=False):
        """Overwrite state tensors (teacher-forced parity tests). With refresh=True the observation, collision flag and
        link distances are recomputed for all envs (obstacle placed at obst_start).  `obst_vel` may be given as the full [9, N]
        snapshot of get_state() or as a [6, N] twist, in which case the per-step displacement (rows 6..8) is re-derived from it."""
        derive = False
        for k, v in state.items():
            v = np.asarray(v)
            if k == "obst_vel" and v.shape[0] == 6:
                # a twist of the caller's own (rows 0..5): rows 6..8, the displacement of one env step under it, are derived state
                self.buf[k][:6].copy_(torch.as_tensor(v, device=self.device).reshape(6, self.num_envs).to(self.buf[k].dtype))
                derive = True
                continue
            t = torch.as_tensor(v, device=self.device).reshape(self.buf[k].shape)
            self.buf[k].copy_(t.to(self.buf[k].dtype))
        if derive:
            _native.check(self.lib.urgym_derive_obstacle_motion(self._h, self._stream()), self._h)
        if "episode_id" in state or "step_count" in state:
            # the library keeps the next episodes of every env prefetched, keyed by episode id: tell it they may no longer match
            _native.check(self.lib.urgym_invalidate_records(self._h), self._h)
        if refresh:
            self._refresh(None)
        self._needs_reset = False

    def probe_closest(self, type_a, par_a, pose_a, type_b, par_b, pose_b, threshold=5.0):
        """Unit probe of the device closest-distance routine (urgym_probe_closest): arrays of queries -> (dist, info)."""
        dev = self.device
        ta = torch.as_tensor(np.asarray(type_a, np.int32), device=dev)
        tb = torch.as_tensor(np.asarray(type_b, np.int32), device=dev)
        n = ta.numel()
        pa = torch.as_tensor(np.asarray(par_a, np.float64).reshape(n, 3), device=dev).contiguous()
        pb = torch.as_tensor(np.asarray(par_b, np.float64).reshape(n, 3), device=dev).contiguous()
        xa = torch.as_tensor(np.asarray(pose_a, np.float64).reshape(n, 7), device=dev).contiguous()
        xb = torch.as_tensor(np.asarray(pose_b, np.float64).reshape(n, 7), device=dev).contiguous()
        out = torch.zeros(n, dtype=torch.float64, device=dev)
        info = torch.zeros(n, dtype=torch.int32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        _native.check(self.lib.urgym_probe_closest(self._h, n, p(ta), p(pa), p(xa), p(tb), p(pb), p(xb), float(threshold), p(out), p(info),
                                                  self._stream()), self._h)
        return out.cpu().numpy(), info.cpu().numpy()

    def probe_pose_distance(self, a6, b6):
        """Unit probe of the device utils.distance / utils.angular_distance (urgym_probe_pose_distance): [n, 6] poses -> [n, 2]."""
        a = torch.as_tensor(np.asarray(a6, np.float64).reshape(-1, 6), device=self.device).contiguous()
        b = torch.as_tensor(np.asarray(b6, np.float64).reshape(-1, 6), device=self.device).contiguous()
        out = torch.zeros((a.shape[0], 2), dtype=torch.float64, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        _native.check(self.lib.urgym_probe_pose_distance(self._h, a.shape[0], p(a), p(b), p(out), self._stream()), self._h)
        return out.cpu().numpy()

    # ------------------------------------------------------------------------------------------------ timing
    def enable_timing(self, on=True, every=1):
        """HIP events around the step launch; every=k times each k-th step only (a pair of events costs the stream ~6 us)."""
        _native.check(self.lib.urgym_enable_timing(self._h, (max(1, int(every)) if on else 0)), self._h)

    def query_timing(self):
        """(avg step-kernel us, avg reset-kernel us, #step launches) since the last query — HIP events on the launch stream."""
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        _native.check(self.lib.urgym_query_timing(self._h, C.byref(a), C.byref(b), C.byref(n)), self._h)
        r = C.c_double()
        _native.check(self.lib.urgym_query_refill_timing(self._h, C.byref(r)), self._h)
        self.last_refill_us = r.value  # overlapped refill of prefetched episode records (0 when that path is off)
        return a.value, b.value, n.value


class _LazyInfo(dict):
    """info dict whose derived entries are computed on first access: a `step()` that nobody inspects launches no extra kernel."""

    def __init__(self, base, lazy):
        super().__init__(base)
        self._lazy = dict(lazy)

    def _materialise(self):
        for k in list(self._lazy):
            dict.__setitem__(self, k, self._lazy.pop(k)())

    def __missing__(self, key):
        if key in self._lazy:
            value = self._lazy.pop(key)()
            dict.__setitem__(self, key, value)
            return value
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + len(self._lazy)


def make_vec(env_id, num_envs=1, **kwargs):
    """Counterpart of ``gymnasium.make(id)`` for the ids the reference registers (UR_gym/__init__.py:19-42)."""
    return UR5ReachVectorEnv(env_id, num_envs=num_envs, **kwargs)
