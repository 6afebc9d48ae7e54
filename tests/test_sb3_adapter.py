"""SB3 VecEnv adapter: exercised with a fake consumer — stable-baselines3 is not installed here."""
import numpy as np
import pytest
import torch

from ur_gym_amd.sb3_adapter import SB3VecEnvAdapter


class FakeVecEnv:
    """Stand-in with the surface of UR5ReachVectorEnv (torch tensors, auto-reset with final_observation)."""

    def __init__(self, n=6):
        self.num_envs = n
        self.single_observation_space = "obs-space"
        self.single_action_space = "act-space"
        self.t = torch.zeros(n, dtype=torch.int64)
        self.closed = False

    def _obs(self):
        o = self.t.float().unsqueeze(1)
        return {"observation": o.repeat(1, 4), "achieved_goal": o.repeat(1, 2), "desired_goal": torch.zeros(self.num_envs, 2)}

    def reset(self, seed=None, options=None):
        self.t[:] = 0
        return self._obs(), {}

    def step(self, actions):
        assert actions.shape == (self.num_envs, 6) and actions.dtype == np.float32
        self.t += 1
        final = self._obs()
        limit = torch.arange(self.num_envs) + 2           # env i finishes every (i+2) steps
        term = (self.t >= limit) & (torch.arange(self.num_envs) % 2 == 0)
        trunc = (self.t >= limit) & (torch.arange(self.num_envs) % 2 == 1)
        rew = torch.full((self.num_envs,), -1.5)
        done = term | trunc
        self.t[done] = 0
        info = {"is_success": term.clone(), "final_observation": final, "_final_observation": done}
        return self._obs(), rew, term, trunc, info

    def close(self):
        self.closed = True


def test_adapter_contract_with_fake_consumer():
    env = FakeVecEnv()
    v = SB3VecEnvAdapter(env)
    obs = v.reset()
    assert set(obs) == {"observation", "achieved_goal", "desired_goal"} and obs["observation"].shape == (6, 4)
    episodes = []
    for k in range(12):
        v.step_async(np.zeros((6, 6)))
        obs, rew, dones, infos = v.step_wait()
        assert rew.dtype == np.float32 and dones.dtype == bool and len(infos) == 6
        for i, info in enumerate(infos):
            if dones[i]:
                # SB3 convention: obs already belongs to the next episode, the last one travels in the info dict
                assert obs["observation"][i, 0] == 0.0
                assert info["terminal_observation"]["observation"][0] == i + 2
                assert info["TimeLimit.truncated"] == (i % 2 == 1)
                assert info["episode"]["l"] == i + 2 and abs(info["episode"]["r"] + 1.5 * (i + 2)) < 1e-6
                episodes.append(i)
            else:
                assert "terminal_observation" not in info
    assert episodes.count(0) == 6 and episodes.count(5) == 1
    assert v.get_attr("num_envs") == [6] * 6 and v.env_is_wrapped(object) == [False] * 6
    v.close()
    assert env.closed


@pytest.mark.gpu
def test_adapter_on_the_hip_environment():
    from ur_gym_amd import make_vec

    env = make_vec("UR5DynReach-v1", num_envs=512, device="cuda:0", seed=3)
    v = SB3VecEnvAdapter(env)
    obs = v.reset()
    assert obs["observation"].shape == (512, 35) and obs["observation"].dtype == np.float32
    rng = np.random.default_rng(0)
    n_done, lengths = 0, []
    for k in range(110):
        obs, rew, dones, infos = v.step(rng.uniform(-1, 1, (512, 6)).astype(np.float32))
        for i in np.nonzero(dones)[0]:
            assert infos[i]["terminal_observation"]["observation"].shape == (35,)
            lengths.append(infos[i]["episode"]["l"])
            n_done += 1
    # every env finished at least once within 110 steps (TimeLimit = 100) and no episode is longer than that
    assert n_done >= 512 and max(lengths) <= 100
    v.close()
