"""stable-baselines3 ``VecEnv`` adapter for the batched HIP environment ("train.py / SAC drops in").

SB3 does not consume gymnasium's VectorEnv; it drives its own ``VecEnv`` ABC (``reset() -> obs``, ``step_async`` /
``step_wait() -> (obs, rewards, dones, infos)`` with ``infos[i]["terminal_observation"]`` and
``infos[i]["TimeLimit.truncated"]`` for finished envs).  ``train.py:39-60`` wraps its single env in ``Monitor`` and
``DummyVecEnv``; this adapter plays both roles for N GPU environments: host numpy views for SB3's replay buffer and
``infos[i]["episode"] = {"r", "l", "t"}`` records like ``Monitor`` / ``VecMonitor`` write to ``monitor.csv``.

stable-baselines3 is optional (it is not installed in the build image): when importable the class derives from its
``VecEnv``; otherwise it is a duck-typed stand-in with the same methods, which is what the tests exercise.
"""
import time

import numpy as np

try:  # pragma: no cover - depends on the environment
    from stable_baselines3.common.vec_env import VecEnv as _Base

    _HAVE_SB3 = True
except Exception:  # pragma: no cover
    _Base = object
    _HAVE_SB3 = False


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


class SB3VecEnvAdapter(_Base):
    """Wraps a ``UR5ReachVectorEnv`` (auto_reset=True).  Observations are dicts of float32 numpy arrays [N, dim]."""

    def __init__(self, env):
        self.env = env
        self.num_envs = env.num_envs
        self.observation_space = env.single_observation_space
        self.action_space = env.single_action_space
        if _HAVE_SB3:
            super().__init__(env.num_envs, env.single_observation_space, env.single_action_space)
        self._actions = None
        self._ep_ret = np.zeros(self.num_envs, dtype=np.float64)
        self._ep_len = np.zeros(self.num_envs, dtype=np.int64)
        self._t0 = time.time()
        self.render_mode = None

    # ------------------------------------------------------------------------------------------------ VecEnv API
    def reset(self):
        obs, _ = self.env.reset()
        self._ep_ret[:] = 0.0
        self._ep_len[:] = 0
        return {k: _np(v).copy() for k, v in obs.items()}

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        obs, rew, term, trunc, info = self.env.step(self._actions)
        obs = {k: _np(v).copy() for k, v in obs.items()}
        rew = _np(rew).astype(np.float32).copy()
        term, trunc = _np(term).astype(bool), _np(trunc).astype(bool)
        dones = term | trunc
        succ = _np(info["is_success"]).astype(bool)
        self._ep_ret += rew
        self._ep_len += 1
        infos = [{"is_success": bool(succ[i])} for i in range(self.num_envs)]
        if dones.any():
            final = {k: _np(v) for k, v in info["final_observation"].items()}
            now = round(time.time() - self._t0, 6)
            for i in np.nonzero(dones)[0]:
                infos[i]["terminal_observation"] = {k: v[i].copy() for k, v in final.items()}
                infos[i]["TimeLimit.truncated"] = bool(trunc[i] and not term[i])
                infos[i]["episode"] = {"r": round(float(self._ep_ret[i]), 6), "l": int(self._ep_len[i]), "t": now}
            self._ep_ret[dones] = 0.0
            self._ep_len[dones] = 0
        return obs, rew, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    def seed(self, seed=None):
        if seed is not None:
            self.env._seed = int(seed)
            self.env._needs_reset = True
        return [seed] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name)] * len(self._indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return [getattr(self.env, method_name)(*args, **kwargs)] * len(self._indices(indices))

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        return [None] * self.num_envs

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)

    # ------------------------------------------------------------------------------------------------ HER support
    def compute_reward(self, achieved_goal, desired_goal, info):
        """Goal-env hook some SB3 components call (core.py:250 exposes task.compute_reward): only the pose-distance
        terms can be re-evaluated from goals alone; the obstacle terms need the environment state."""
        raise NotImplementedError("relabelled rewards need link distances; use the rewards returned by step()")
