"""Closed-loop replay of the reference's shipped SAC actors: the only END-TO-END pin the reference offers.

The actors were trained by the reference authors in their PyBullet environments; ``Trained_Models/*/best*.txt`` records
what they achieve there (tests/golden/actors/reference_results.json).  A wrong frame, Euler convention, observation
layout, obstacle motion or collision rule collapses these numbers, so reproducing them is strong evidence that the
restatement means the same thing as the reference.  Measured with the oracle (CPU) and the HIP path (GPU):

    UR5OriReach-v1   reference 97.28 %  /  this build 97.2 %    (5250 trials: goal grid x 5 orientations)
    UR5DynReach-v1   reference 96.24 %  /  this build 96.0 %    (3675 trials: goal grid x 5 obstacle draws)
    UR5StaReach-v1   reference 89.44 %  /  this build ~60 %     (5000 resets)  <- KNOWN GAP
    UR5ObsReach-v1   reference 95.90 %  /  this build ~65 %     (5000 resets)  <- KNOWN GAP, see DESIGN.md §3:
        both older checkpoints drive the arm to a FIXED POINT (zero action) 5-9 cm / 0.04-0.16 rad from the goal, i.e.
        just outside the success zone (0.05 m, 0.0873 rad) of the current code in a third of the trials; with thresholds
        (0.1 m, 0.2 rad) the Sta replay gives 91 % in 10 steps.  Obs additionally ends 28 % of its episodes in table
        contacts of forearm / wrist-1 at goals below z = 0 (its target is a collidable sphere 2 cm above a collidable
        table: contact dynamics of stepSimulation, SURVEY.md §7 H4-ii, are not modelled).  These two checkpoints appear
        to predate the present thresholds / scene of the reference; they are reported, not used as pins.

The per-trial test points of the reference were drawn from unseeded RNGs and never saved, so only the aggregates are
comparable; the tolerances below are a few standard errors of a binomial proportion.
"""
import json
import os

import numpy as np
import pytest

from ur_gym_amd import _abi
from ur_gym_amd.evaluation import DeterministicActor, constrained_euler, goal_grid, run_closed_loop

HERE = os.path.dirname(os.path.abspath(__file__))
ACTORS = os.path.join(HERE, "golden", "actors")
with open(os.path.join(ACTORS, "reference_results.json")) as _f:
    REF = json.load(_f)

ORI_LOW, ORI_HIGH = np.array([0.3, -0.5, 0.0]), np.array([0.75, 0.5, 0.2])   # reach.py:151-152
DYN_LOW, DYN_HIGH = np.array([0.4, -0.5, 0.0]), np.array([0.75, 0.5, 0.2])   # reach.py:584-585


class OracleBackend:
    def __init__(self, env):
        self.env, self.num_envs = env, env.num_envs

    def observe(self):
        b = self.env.buf
        return b["achieved_goal"], b["desired_goal"], b["observation"]

    def step(self, actions):
        self.env.step(actions)
        b = self.env.buf
        return b["reward"].astype(np.float64), b["terminated"].copy(), b["is_success"].copy()


def thin(points, keep):
    """Deterministic subset (every k-th trial) to keep the CPU suite short."""
    return points if keep >= len(points) else points[:: max(1, len(points) // keep)][:keep]


def dyn_points(draw_states, clearance, n_trials, rng):
    """utils/generate.generate_dyn + ReachDyn.reset_generate (reach.py:685-700): grid goal, sampled goal orientation and
    obstacle start/end with travel >= 0.3 and target <-> obstacle(end) clearance >= 0.1."""
    grid = goal_grid(DYN_LOW, DYN_HIGH)
    grid = thin(grid, n_trials)
    goal_rpy, start, end = draw_states(3 * len(grid))
    pts, j = [], 0
    cand_g = np.concatenate([np.repeat(grid, 3, axis=0), goal_rpy[: 3 * len(grid)]], axis=1)
    ok = clearance(cand_g, end[: 3 * len(grid)]) >= 0.1
    for i in range(len(grid)):
        choices = [k for k in range(3 * i, 3 * i + 3) if ok[k]]
        k = choices[0] if choices else 3 * i
        pts.append(np.r_[cand_g[k], start[k], end[k]])
    return np.array(pts)


# ------------------------------------------------------------------------------------------------ CPU (oracle)
def test_ori_actor_closed_loop_oracle(oracle):
    rng = np.random.default_rng(0)
    grid = thin(goal_grid(ORI_LOW, ORI_HIGH), 1750)
    pts = np.concatenate([grid, constrained_euler(rng, len(grid))], axis=1)
    env = oracle.OracleEnv(_abi.ENV_ORI, len(pts), threads=8, auto_reset=0)
    env.reset(seed=1)
    env.buf["goal"][:] = pts.T      # ReachOri.set_goal (reach.py:202-204)
    env.refresh()
    res = run_closed_loop(OracleBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_ori.npz")))
    print("Ori closed loop (oracle):", {k: res[k] for k in ("success_rate_percent", "mean_episode_reward", "mean_last_step_index")}, "reference:", REF["ori"])
    assert abs(res["success_rate_percent"] - REF["ori"]["success_rate_percent"]) < 2.5
    assert abs(res["mean_episode_reward"] - REF["ori"]["mean_episode_reward"]) < 25.0
    assert abs(res["mean_last_step_index"] - REF["ori"]["mean_last_step_index"]) < 1.5
    env.close()


def test_dyn_actor_closed_loop_oracle(oracle):
    rng = np.random.default_rng(0)

    def draw(n):
        gen = oracle.OracleEnv(_abi.ENV_DYN, n, threads=8, auto_reset=0, min_travel=0.3)
        gen.reset(seed=3)
        return gen.buf["goal"][3:].T.copy(), gen.buf["obst_start"].T.copy(), gen.buf["obst_end"].T.copy()

    def clearance(goal6, end6):
        return np.array([oracle.closest(oracle.BOX, [0.025] * 3, np.r_[g[:3], oracle.quat_from_euler(g[3:])], oracle.CYLZ, [0.05, 0.4],
                                        np.r_[e[:3], oracle.quat_from_euler(e[3:])])["distance"] for g, e in zip(goal6, end6)])

    pts = dyn_points(draw, clearance, 1200, rng)
    env = oracle.OracleEnv(_abi.ENV_DYN, len(pts), threads=8, auto_reset=0)
    env.reset(seed=4)
    env.buf["goal"][:] = pts[:, :6].T
    env.buf["obst_start"][:] = pts[:, 6:12].T
    env.buf["obst_end"][:] = pts[:, 12:].T
    env.refresh()  # ReachDyn.set_goal_and_obstacle (reach.py:702-713)
    res = run_closed_loop(OracleBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_dyn.npz")))
    print("Dyn closed loop (oracle):", {k: res[k] for k in ("success_rate_percent", "mean_episode_reward", "mean_last_step_index")}, "reference:", REF["dyn"])
    assert abs(res["success_rate_percent"] - REF["dyn"]["success_rate_percent"]) < 3.0
    assert abs(res["mean_episode_reward"] - REF["dyn"]["mean_episode_reward"]) < 30.0
    assert abs(res["mean_last_step_index"] - REF["dyn"]["mean_last_step_index"]) < 1.5
    env.close()


def test_sta_actor_closed_loop_oracle(oracle):
    """UR5StaReach-v1: generate_sta = 5000 x task.reset() (utils/generate.py:47-56)."""
    env = oracle.OracleEnv(_abi.ENV_STA, 1200, threads=8, auto_reset=0)
    env.reset(seed=5)
    res = run_closed_loop(OracleBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_sta.npz")))
    print("Sta closed loop (oracle):", {k: res[k] for k in ("success_rate_percent", "mean_episode_reward", "mean_last_step_index")}, "reference:", REF["sta"])
    # documented gap (module docstring): reported, loosely bounded
    assert 45.0 < res["success_rate_percent"] <= 100.0
    env.close()
    # with the looser thresholds the policy's fixed point falls inside the success zone
    env = oracle.OracleEnv(_abi.ENV_STA, 600, threads=8, auto_reset=0, distance_threshold=0.1, ori_threshold=0.2)
    env.reset(seed=5)
    loose = run_closed_loop(OracleBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_sta.npz")))
    print("Sta closed loop (oracle, thresholds 0.1 m / 0.2 rad):", loose["success_rate_percent"], loose["mean_last_step_index"])
    assert loose["success_rate_percent"] > 85.0
    env.close()


def test_obs_actor_closed_loop_oracle_known_gap(oracle):
    env = oracle.OracleEnv(_abi.ENV_OBS, 1000, threads=8, auto_reset=0)
    env.reset(seed=2)  # generate_obs = 5000 x task.reset() (utils/generate.py:91-102)
    res = run_closed_loop(OracleBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_obs.npz")))
    print("Obs closed loop (oracle):", {k: res[k] for k in ("success_rate_percent", "mean_episode_reward", "mean_last_step_index")}, "reference:", REF["obs"])
    # documented gap (module docstring): the actor still reaches most goals, but well below the reference's 95.9 %
    assert 50.0 < res["success_rate_percent"] <= 100.0
    env.close()


# ------------------------------------------------------------------------------------------------ GPU (HIP path)
@pytest.mark.gpu
def test_closed_loop_hip_full_protocol():
    import torch

    from ur_gym_amd import make_vec
    from ur_gym_amd.evaluation import HipBackend

    rng = np.random.default_rng(0)
    out = {}
    # Ori: 5250 trials
    grid = goal_grid(ORI_LOW, ORI_HIGH)
    pts = np.concatenate([grid, constrained_euler(rng, len(grid))], axis=1)
    env = make_vec("UR5OriReach-v1", num_envs=len(pts), device="cuda:0", seed=1, auto_reset=False)
    env.reset(seed=1)
    env.set_goal(np.arange(len(pts)), pts)
    out["ori"] = run_closed_loop(HipBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_ori.npz")))
    env.close()

    # Dyn: 3675 trials
    def draw(n):
        gen = make_vec("UR5DynReach-v1", num_envs=n, device="cuda:0", seed=3, auto_reset=False, min_travel=0.3)
        gen.reset(seed=3)
        torch.cuda.synchronize()
        st = gen.get_state()
        gen.close()
        return st["goal"][3:].T.copy(), st["obst_start"].T.copy(), st["obst_end"].T.copy()

    probe_env = make_vec("UR5DynReach-v1", num_envs=64, device="cuda:0", seed=0)

    def clearance(goal6, end6):
        from scipy.spatial.transform import Rotation as Rot

        n = len(goal6)
        qa = Rot.from_euler("xyz", goal6[:, 3:]).as_quat()
        qb = Rot.from_euler("xyz", end6[:, 3:]).as_quat()
        d, _ = probe_env.probe_closest(np.full(n, 2), np.tile([0.025] * 3, (n, 1)), np.c_[goal6[:, :3], qa], np.full(n, 1),
                                       np.tile([0.05, 0.4, 0.0], (n, 1)), np.c_[end6[:, :3], qb])
        return d

    pts = dyn_points(draw, clearance, 10 ** 9, rng)
    probe_env.close()
    env = make_vec("UR5DynReach-v1", num_envs=len(pts), device="cuda:0", seed=4, auto_reset=False)
    env.reset(seed=4)
    env.set_goal_and_obstacle(np.arange(len(pts)), pts)
    out["dyn"] = run_closed_loop(HipBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_dyn.npz")))
    env.close()

    # Sta: 5000 resets
    env = make_vec("UR5StaReach-v1", num_envs=5000, device="cuda:0", seed=5, auto_reset=False)
    env.reset(seed=5)
    out["sta"] = run_closed_loop(HipBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_sta.npz")))
    env.close()

    # Obs: 5000 resets (known gap, reported only)
    env = make_vec("UR5ObsReach-v1", num_envs=5000, device="cuda:0", seed=2, auto_reset=False)
    env.reset(seed=2)
    out["obs"] = run_closed_loop(HipBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_obs.npz")))
    env.close()

    for k in ("ori", "dyn", "sta", "obs"):
        r = out[k]
        print(f"{k} closed loop (HIP): success {r['success_rate_percent']:.2f}%  reward {r['mean_episode_reward']:.2f}  "
              f"last step {r['mean_last_step_index']:.2f}   reference: {REF[k]}")
    assert len(out["ori"]["success"]) == REF["ori"]["trials"] and len(out["dyn"]["success"]) == REF["dyn"]["trials"]
    assert abs(out["ori"]["success_rate_percent"] - REF["ori"]["success_rate_percent"]) < 1.5
    assert abs(out["dyn"]["success_rate_percent"] - REF["dyn"]["success_rate_percent"]) < 2.0
    assert abs(out["ori"]["mean_last_step_index"] - REF["ori"]["mean_last_step_index"]) < 1.0
    assert abs(out["dyn"]["mean_last_step_index"] - REF["dyn"]["mean_last_step_index"]) < 1.0
    assert 45.0 < out["sta"]["success_rate_percent"] <= 100.0  # known gap, reported
    assert 50.0 < out["obs"]["success_rate_percent"] <= 100.0  # known gap, reported
