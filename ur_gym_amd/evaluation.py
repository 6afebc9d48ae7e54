"""Closed-loop evaluation harness: counterpart of the reference's ``model_test.py:26-61`` + ``utils/generate.py:23-102``.

The reference evaluates a trained SAC actor by (1) generating test points per env (a goal grid with sampled
orientations / obstacles, or plain resets), (2) ``env.reset(); env.task.set_goal[_and_obstacle](point)``, (3) rolling
the deterministic policy for at most 100 steps and recording episode reward, ``info['is_success']`` and the index of
the last step.  Here all trials run in parallel — one environment per trial — on either backend that offers the
VectorEnv verbs (the HIP environment of this package, or, in tests, the CPU oracle through a thin adapter).

The actor is re-implemented from the checkpoint's tensors (SB3 ``MultiInputPolicy``: features = concat of the Dict
observation in sorted key order achieved_goal | desired_goal | observation; ``latent_pi`` = Linear-ReLU-Linear-ReLU;
``mu`` = Linear; deterministic action = tanh(mu)); stable-baselines3 itself is not needed.
"""
import numpy as np


class DeterministicActor:
    """tanh(mu(latent_pi(x))) from the arrays exported by tests/golden/gen_actor_fixtures.py (or any SB3 SAC policy.pth)."""

    def __init__(self, weights):
        self.w = {k: np.asarray(v, dtype=np.float32) for k, v in dict(weights).items()}
        self.in_features = self.w["latent_pi_0_weight"].shape[1]

    @classmethod
    def load(cls, npz_path):
        return cls(np.load(npz_path))

    def __call__(self, achieved_goal, desired_goal, observation):
        x = np.concatenate([achieved_goal, desired_goal, observation], axis=1).astype(np.float32)
        assert x.shape[1] == self.in_features, (x.shape, self.in_features)
        h = np.maximum(x @ self.w["latent_pi_0_weight"].T + self.w["latent_pi_0_bias"], 0.0)
        h = np.maximum(h @ self.w["latent_pi_2_weight"].T + self.w["latent_pi_2_bias"], 0.0)
        return np.tanh(h @ self.w["mu_weight"].T + self.w["mu_bias"]).astype(np.float32)


def goal_grid(low, high, step=0.05, repeats=5):
    """utils/generate.py:29-43, 63-80: every grid node of the goal range, `repeats` times (float arithmetic as there)."""
    n = [int((high[i] - low[i]) / step) + 1 for i in range(3)]
    pts = [(low[0] + i * step, low[1] + j * step, low[2] + k * step)
           for i in range(n[0]) for j in range(n[1]) for k in range(n[2]) for _ in range(repeats)]
    return np.array(pts, dtype=np.float64)


def constrained_euler(rng, n):
    """utils.sample_euler_constrained (utils.py:81-86) for n goals."""
    return np.stack([np.deg2rad(-rng.uniform(90, 180, n)), np.zeros(n), np.deg2rad(-rng.uniform(0, 180, n))], axis=1)


def run_closed_loop(backend, actor, max_steps=100):
    """model_test.run_test (model_test.py:26-61) for all envs of `backend` at once.

    backend: object with ``num_envs``, ``observe() -> (achieved, desired, observation)`` numpy arrays and
    ``step(actions) -> (reward, terminated, is_success)`` numpy arrays; auto-reset must be OFF.
    Returns dict(success_rate_percent, mean_episode_reward, mean_last_step_index, per-trial arrays).
    """
    n = backend.num_envs
    done = np.zeros(n, bool)
    success = np.zeros(n, bool)
    reward = np.zeros(n)
    last = np.zeros(n)
    for t in range(max_steps):
        a = actor(*backend.observe())
        r, term, succ = backend.step(a)
        live = ~done
        reward[live] += r[live]
        fin = live & (term.astype(bool) | (t == max_steps - 1))  # model_test.py:46: `if steps == 99 or terminated`
        success[fin] = succ[fin].astype(bool)
        last[fin] = t
        done |= fin
        if done.all():
            break
    return {"success_rate_percent": 100.0 * success.mean(), "mean_episode_reward": float(reward.mean()),
            "mean_last_step_index": float(last.mean()), "success": success, "reward": reward, "last_step": last}


class HipBackend:
    """Adapter of UR5ReachVectorEnv (created with auto_reset=False) for run_closed_loop."""

    def __init__(self, env):
        import torch

        self.env, self.torch = env, torch
        self.num_envs = env.num_envs

    def observe(self):
        b = self.env.buf
        return b["achieved_goal"].cpu().numpy(), b["desired_goal"].cpu().numpy(), b["observation"].cpu().numpy()

    def step(self, actions):
        obs, rew, term, trunc, info = self.env.step(self.torch.from_numpy(actions).to(self.env.device))
        return rew.cpu().numpy().astype(np.float64), term.cpu().numpy(), info["is_success"].cpu().numpy()
