"""Closed-loop replay of the reference's shipped SAC actors: the END-TO-END pin the reference offers.

The actors were trained by the reference authors in their PyBullet environments; ``Trained_Models/*/best*.txt`` records,
trial by trial, what they achieve there (tests/golden/actors/reference_results.json: aggregates AND the per-trial
statistics -- early failures, time-outs, percentiles of the last step, mean reward of the successful trials).  A wrong
frame, Euler convention, observation layout, obstacle motion or collision rule collapses these numbers.

    env               reference (5250 / 5000 / 5000 / 3675 trials)     this build (oracle, CPU; the HIP path agrees)
    UR5OriReach-v1    97.28 %   last step 8.35                          97.2 %   8.4
    UR5DynReach-v1    96.24 %   last step 7.70                          96.0 %   8.1
    UR5ObsReach-v1    95.90 %   9.56   early-fail 2.26 %  t/o 1.84 %    96.8 %   9.0   1.6 %  1.6 %     (link_dist_scope = WORKBENCH)
    UR5StaReach-v1    89.44 %  13.52   early-fail 3.64 %  t/o 6.92 %    88.4 %  12.9   5.4 %  6.2 %     (link_dist_scope = WORKBENCH)

ATTRIBUTION OF ROUND 1'S OBS / STA GAP (65 % / 61 %).  The Obs and Sta checkpoints date from September 2023, Dyn from May
2024 (system_info.txt inside the zips).  The observations stored with the checkpoints
(tests/golden/reference_observations.json, tests/test_reference_pins.py) show to 1e-7 m that in September 2023
``PyBullet.get_link_distances`` returned, per link, the MINIMUM distance to obstacle, table and track -- what its docstring
still says ("the distance between workbench, obstacle and UR5", pyb_setup.py:440) -- whereas the code as it stands
(pyb_setup.py:449-454), and the Dyn checkpoint's observations, measure the obstacle only.  Fed with today's link_dist the two
old actors are out of distribution: the upper-arm slot reads 0.3-0.7 m instead of the constant ~0.1015 m to the track they
were trained on, and they stall just outside the success zone or graze the table.  With urgym_config.link_dist_scope =
URGYM_LINK_DIST_WORKBENCH every per-trial statistic of the reference is reproduced within sampling error (ablation:
tools/closed_loop_ablation.py); thresholds, table/track checks and self-collision stay as in the current code.

The per-trial test points of the reference were drawn from unseeded RNGs and never saved, so only statistics are
comparable; every tolerance below is three standard errors of the difference of the two estimates (check_against_reference).
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
    check_against_reference("ori", res, len(pts), reward=True)
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
    check_against_reference("dyn", res, len(pts), reward=True)
    env.close()


def trial_stats(res):
    s, l = res["success"], res["last_step"]
    return {"early_fail_percent": 100.0 * float((~s & (l < 99)).mean()), "timeout_percent": 100.0 * float((l >= 99).mean()),
            "success_last_step_p50": float(np.percentile(l[s], 50)), "success_last_step_p95": float(np.percentile(l[s], 95))}


TRIALS = np.load(os.path.join(ACTORS, "reference_trials.npz"))  # per-trial rows of best.txt / best_modeltest_result.txt (reward, success, last step)
Z = 3.0  # every bound below is Z standard errors of the difference of the two estimates (both are samples)


def se_of_difference(sd_a, n_a, sd_b, n_b):
    return float(np.sqrt(sd_a ** 2 / max(n_a, 1) + sd_b ** 2 / max(n_b, 1)))


def check_against_reference(name, res, n, reward=False):
    """Every statistic the reference's per-trial rows carry, each within Z = 3 standard errors of the DIFFERENCE of the two sample
    estimates (no hand-picked tolerances): success rate, early failures, time-outs (binomial); mean last step (sample standard
    deviations of both sides); median / 95 % point of the successful trials' last step (integers: +-1).  reward=True -- the two
    checkpoints trained with the reference's present code, Ori and Dyn, for which the reward formula is pinned end to end: also the
    mean episode reward, the mean reward of the successful AND of the failed trials, and the 5 / 50 / 95 % quantiles of the per-trial
    episode reward (rank test: the share of our trials below the reference's quantile value against q)."""
    ref, st = REF[name], trial_stats(res)
    rew, ok, last = np.asarray(res["reward"], dtype=np.float64), np.asarray(res["success"], dtype=bool), np.asarray(res["last_step"], dtype=np.float64)
    m = ref["trials"]
    print(f"{name} closed loop:", {k: round(res[k], 2) for k in ("success_rate_percent", "mean_episode_reward", "mean_last_step_index")}, st,
          "reference:", {k: v for k, v in ref.items() if not isinstance(v, dict)})

    def binomial(got_percent, ref_percent, label):
        q = min(max(ref_percent, 0.5), 99.5) / 100.0
        se = 100.0 * np.sqrt(q * (1 - q) * (1.0 / n + 1.0 / m))
        assert abs(got_percent - ref_percent) < Z * se, (name, label, got_percent, ref_percent, se)

    def mean(got, sd_got, n_got, ref_mean, sd_ref, n_ref, label):
        se = se_of_difference(sd_got, n_got, sd_ref, n_ref)
        assert abs(got - ref_mean) < Z * se, (name, label, got, ref_mean, se)

    binomial(res["success_rate_percent"], ref["success_rate_percent"], "success rate")
    binomial(st["early_fail_percent"], ref["early_fail_percent"], "early failures")
    binomial(st["timeout_percent"], ref["timeout_percent"], "time-outs")
    mean(float(last.mean()), float(last.std(ddof=1)), n, ref["mean_last_step_index"], ref["sd_last_step_index"], m, "mean last step")
    assert abs(st["success_last_step_p50"] - ref["success_last_step_p50"]) <= 1.0
    assert abs(st["success_last_step_p95"] - ref["success_last_step_p95"]) <= 1.0
    if reward:
        mean(float(rew.mean()), float(rew.std(ddof=1)), n, ref["mean_episode_reward"], ref["sd_episode_reward"], m, "mean episode reward")
        mean(float(rew[ok].mean()), float(rew[ok].std(ddof=1)), int(ok.sum()), ref["mean_success_reward"], ref["sd_success_reward"],
             m - ref["failures"], "mean reward of the successful trials")
        if (~ok).sum() >= 10:
            mean(float(rew[~ok].mean()), float(rew[~ok].std(ddof=1)), int((~ok).sum()), ref["mean_failure_reward"], ref["sd_failure_reward"],
                 ref["failures"], "mean reward of the failed trials")
        for q in (5, 50, 95):
            x = ref["episode_reward_quantiles"][str(q)]
            share = float((rew <= x).mean())
            se = np.sqrt(q / 100.0 * (1 - q / 100.0) * (1.0 / n + 1.0 / m))
            assert abs(share - q / 100.0) < Z * se, (name, f"{q} % quantile of the episode reward", x, share, se)
        # the whole distribution once more: two-sample Kolmogorov-Smirnov distance against the reference's per-trial rewards
        a, b = np.sort(rew), np.sort(TRIALS[f"{name}_reward"])
        grid = np.concatenate([a, b])
        ks = float(np.abs(np.searchsorted(a, grid, side="right") / len(a) - np.searchsorted(b, grid, side="right") / len(b)).max())
        assert ks < 1.95 * np.sqrt((len(a) + len(b)) / (len(a) * len(b))), (name, "KS distance of the episode rewards", ks)  # alpha = 0.001


def test_sta_actor_closed_loop_oracle(oracle):
    """UR5StaReach-v1: generate_sta = 5000 x task.reset() (utils/generate.py:47-56); Sep-2023 checkpoint."""
    n = 1500
    env = oracle.OracleEnv(_abi.ENV_STA, n, threads=8, auto_reset=0, link_dist_scope=_abi.LINK_DIST_WORKBENCH)
    env.reset(seed=5)
    res = run_closed_loop(OracleBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_sta.npz")))
    check_against_reference("sta", res, n)
    env.close()


def test_obs_actor_closed_loop_oracle(oracle):
    """UR5ObsReach-v1: generate_obs = 5000 x task.reset() (utils/generate.py:91-102); Sep-2023 checkpoint."""
    n = 1500
    env = oracle.OracleEnv(_abi.ENV_OBS, n, threads=8, auto_reset=0, link_dist_scope=_abi.LINK_DIST_WORKBENCH)
    env.reset(seed=2)
    res = run_closed_loop(OracleBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_obs.npz")))
    check_against_reference("obs", res, n)
    env.close()


def test_old_checkpoints_fail_with_todays_link_dist(oracle):
    """The negative control of the attribution: with the obstacle-only link_dist of the present code the two Sep-2023 actors
    lose a third of their trials (what round 1 measured and could not explain)."""
    for name, kind, seed in (("obs", _abi.ENV_OBS, 2), ("sta", _abi.ENV_STA, 5)):
        env = oracle.OracleEnv(kind, 500, threads=8, auto_reset=0, link_dist_scope=_abi.LINK_DIST_OBSTACLE)
        env.reset(seed=seed)
        res = run_closed_loop(OracleBackend(env), DeterministicActor.load(os.path.join(ACTORS, f"actor_{name}.npz")))
        assert res["success_rate_percent"] < REF[name]["success_rate_percent"] - 15.0, (name, res["success_rate_percent"])
        env.close()


def test_reward_formula_of_the_2023_checkpoints(oracle):
    """What reward produced the per-trial numbers the reference ships for its two Sep-2023 checkpoints?  One replay per env records the
    terms of the reward per step; the trials are then priced under the code as it stands and under the 2023 hypothesis
    (tools/reward_hypotheses.py prices a whole family: profiles/r3/reward_hypotheses.txt).
      Obs: the code as it stands (reach.py:356-374) reproduces the distribution of the successful trials' rewards.
      Sta: the code as it stands (reach.py:543-573: early returns, -70 d - 30 theta, link weights [8, 2.4, 1.2, 1.2, 0.2] / 13 * 50)
           does NOT (mean -127 vs -154.7); the additive form of Ori / Obs with -70 d - 30 theta and the Obs link weight 100 does,
           to a Kolmogorov-Smirnov distance inside the alpha = 0.001 critical value and the means within 3 standard errors."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import reward_hypotheses as rh

    def compare(rew, ok, name):
        ref, ref_ok = TRIALS[f"{name}_reward"], TRIALS[f"{name}_success"].astype(bool)
        a, b = np.sort(rew[ok]), np.sort(ref[ref_ok])
        grid = np.concatenate([a, b])
        ks = float(np.abs(np.searchsorted(a, grid, side="right") / len(a) - np.searchsorted(b, grid, side="right") / len(b)).max())
        crit = 1.95 * np.sqrt((len(a) + len(b)) / (len(a) * len(b)))
        z = abs(rew[ok].mean() - ref[ref_ok].mean()) / se_of_difference(rew[ok].std(ddof=1), ok.sum(), ref[ref_ok].std(ddof=1), ref_ok.sum())
        return ks, crit, z

    n = 1500
    log, ok, _ = rh.replay("obs", _abi.ENV_OBS, n, 2)
    today = sum(np.where(s["live"], s["r_now"], 0.0) for s in log)
    ks, crit, z = compare(today, ok, "obs")
    assert ks < crit and z < Z, ("obs, the code as it stands", ks, crit, z)

    log, ok, _ = rh.replay("sta", _abi.ENV_STA, n, 5)
    today = sum(np.where(s["live"], s["r_now"], 0.0) for s in log)
    ks, crit, z = compare(today, ok, "sta")
    assert ks > crit and z > 2 * Z, ("sta, the code as it stands, should NOT reproduce the 2023 rewards", ks, crit, z)
    # (the replayed `early / -70 d - 30 theta / Dyn weights` pricing IS the code as it stands: the tool's bookkeeping is right)
    same = rh.price(log, n, "early", -70.0, -30.0, rh.W_DYN)
    assert np.abs(same - today).max() < 1e-3
    y2023 = rh.price(log, n, "additive", -70.0, -30.0, np.full(5, 100.0))
    ks, crit, z = compare(y2023, ok, "sta")
    assert ks < crit and z < Z, ("sta, additive form with the Obs link weight", ks, crit, z)
    # the failed trials too (mean within 3 standard errors)
    ref, ref_ok = TRIALS["sta_reward"], TRIALS["sta_success"].astype(bool)
    zf = abs(y2023[~ok].mean() - ref[~ref_ok].mean()) / se_of_difference(y2023[~ok].std(ddof=1), (~ok).sum(), ref[~ref_ok].std(ddof=1), (~ref_ok).sum())
    assert zf < Z, ("sta failed trials", y2023[~ok].mean(), ref[~ref_ok].mean(), zf)


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
    env = make_vec("UR5StaReach-v1", num_envs=5000, device="cuda:0", seed=5, auto_reset=False, link_dist_scope=_abi.LINK_DIST_WORKBENCH)
    env.reset(seed=5)
    out["sta"] = run_closed_loop(HipBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_sta.npz")))
    env.close()

    # Obs: 5000 resets
    env = make_vec("UR5ObsReach-v1", num_envs=5000, device="cuda:0", seed=2, auto_reset=False, link_dist_scope=_abi.LINK_DIST_WORKBENCH)
    env.reset(seed=2)
    out["obs"] = run_closed_loop(HipBackend(env), DeterministicActor.load(os.path.join(ACTORS, "actor_obs.npz")))
    env.close()

    for k in ("ori", "dyn", "sta", "obs"):
        r = out[k]
        print(f"{k} closed loop (HIP): success {r['success_rate_percent']:.2f}%  reward {r['mean_episode_reward']:.2f}  "
              f"last step {r['mean_last_step_index']:.2f}   reference: {REF[k]}")
    assert len(out["ori"]["success"]) == REF["ori"]["trials"] and len(out["dyn"]["success"]) == REF["dyn"]["trials"]
    check_against_reference("ori", out["ori"], REF["ori"]["trials"], reward=True)
    check_against_reference("dyn", out["dyn"], REF["dyn"]["trials"], reward=True)
    check_against_reference("sta", out["sta"], 5000)
    check_against_reference("obs", out["obs"], 5000)
