#!/usr/bin/env python3
"""Which reward did the reference's Sep-2023 checkpoints (UR5ObsReach-v1, UR5StaReach-v1) see when their best.txt was written?

VERDICT r2 item 6: with link_dist_scope = WORKBENCH the two old actors reproduce the reference's success / failure / episode-length
statistics, but not its mean reward of the successful trials (Obs -177 vs -183.0, Sta -128 vs -154.7: the reward code of 2023 is
not the code as it stands).  This replays the actors on the CPU oracle, records the TERMS of the reward per step (position distance d,
orientation distance theta, link distances before / after, success, collision) and prices every trial under a family of plausible
2023 formulas; the trajectories do not depend on the reward, so one replay serves all hypotheses.

    python tools/reward_hypotheses.py [--trials 3000] > profiles/r3/reward_hypotheses.txt
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from ur_gym_amd import _abi  # noqa: E402
from ur_gym_amd.evaluation import DeterministicActor  # noqa: E402

ACTORS = os.path.join(ROOT, "tests", "golden", "actors")
REF = json.load(open(os.path.join(ACTORS, "reference_results.json")))
TRIALS = np.load(os.path.join(ACTORS, "reference_trials.npz"))
W_DYN = np.array([8, 2.4, 1.2, 1.2, 0.2]) / 13 * 50   # reach.py:594-598


def replay(name, kind, n, seed):
    env = ob.OracleEnv(kind, n, threads=8, auto_reset=0, link_dist_scope=_abi.LINK_DIST_WORKBENCH)
    env.reset(seed=seed)
    actor = DeterministicActor.load(os.path.join(ACTORS, f"actor_{name}.npz"))
    done = np.zeros(n, bool)
    log = []  # per step: dict of arrays over the envs still running
    last = np.zeros(n, int)
    succ_final = np.zeros(n, bool)
    for t in range(100):
        b = env.buf
        ld_old = b["link_dist"].copy()
        a = actor(b["achieved_goal"], b["desired_goal"], b["observation"])
        env.step(a)
        b = env.buf
        ach, goal = b["achieved_goal"].astype(np.float64), b["desired_goal"].astype(np.float64)
        d = np.linalg.norm(ach[:, :3] - goal[:, :3], axis=1)
        th = np.zeros(n)
        if kind != _abi.ENV_OBS:
            th = np.array([ob.angular_distance(ach[i], goal[i]) for i in range(n)])  # (the probe takes the 6-vectors: utils.angular_distance reads x[3:6])
        # link distances after this step (WORKBENCH scope), also where the present code returned before updating them
        ld_new = b["link_dist"].copy()
        term = b["terminated"].astype(bool)
        if kind != _abi.ENV_OBS:
            for i in np.nonzero(term & ~done)[0]:
                pose = np.r_[b["obst_pos"][:, i], b["obst_quat"][:, i]]
                ld_new[:, i] = ob.query(b["q"][:, i], pose, scope=_abi.LINK_DIST_WORKBENCH)[0]
        live = ~done
        log.append({"live": live.copy(), "d": d, "th": th, "ld_old": ld_old, "ld_new": ld_new, "succ": b["is_success"].astype(bool) | (term & ~b["collision"].astype(bool)),
                    "coll": b["collision"].astype(bool), "r_now": b["reward"].astype(np.float64).copy()})
        fin = live & (term | (t == 99))
        succ_final[fin] = b["is_success"][fin].astype(bool)
        last[fin] = t
        done |= fin
        if done.all():
            break
    env.close()
    return log, succ_final, last


def price(log, n, form, w_d, w_th, w_link, near=0.2):
    """form 'additive' (reach.py:356-374 today's Obs): 200 succ - 500 coll + w_d d + w_th theta + link term, always;
    form 'early' (reach.py:764-785 today's Dyn / Sta): collision -> -500, success -> +200, else the shaped terms."""
    total = np.zeros(n)
    for s in log:
        link = ((s["ld_new"] < near) * (w_link[:, None] * (s["ld_new"] - s["ld_old"]))).sum(axis=0)
        shaped = w_d * s["d"] + w_th * s["th"] + link
        if form == "additive":
            r = 200.0 * s["succ"] - 500.0 * s["coll"] + shaped
        else:
            r = np.where(s["coll"], -500.0, np.where(s["succ"], 200.0, shaped))
        total += np.where(s["live"], r, 0.0)
    return total


def report(name, kind, n, seed):
    log, succ, last = replay(name, kind, n, seed)
    ref = REF[name]
    ref_rew, ref_ok = TRIALS[f"{name}_reward"], TRIALS[f"{name}_success"].astype(bool)
    now = np.zeros(n)
    for s in log:
        now += np.where(s["live"], s["r_now"], 0.0)
    print(f"== {name}: {n} trials, success {100 * succ.mean():.2f} % (reference {ref['success_rate_percent']:.2f} %), mean last step {last.mean():.2f} ({ref['mean_last_step_index']:.2f})")
    print(f"   reference: mean reward of the successful trials {ref['mean_success_reward']:.1f}, of the failed ones {ref['mean_failure_reward']:.1f}, all {ref['mean_episode_reward']:.1f}")
    print(f"   {'hypothesis':78s} {'success':>9s} {'failed':>9s} {'all':>9s} {'KS vs reference (successful trials)':>36s}")
    hyps = [("the code as it stands (oracle's own reward)", None)]
    w100, w0 = np.full(5, 100.0), np.zeros(5)
    for form in ("additive", "early"):
        for (dn, wd, wt) in (("-100 d", -100.0, 0.0), ("-70 d - 30 theta", -70.0, -30.0)):
            if kind == _abi.ENV_OBS and wt != 0.0:
                continue  # Obs has no orientation goal
            for (ln, wl) in (("link weight 100", w100), ("Dyn link weights", W_DYN), ("no link term", w0)):
                hyps.append((f"{form:8s} | {dn:16s} | {ln}", (form, wd, wt, wl)))
    for label, h in hyps:
        tot = now if h is None else price(log, n, *h)
        a, b = np.sort(tot[succ]), np.sort(ref_rew[ref_ok])
        grid = np.concatenate([a, b])
        ks = float(np.abs(np.searchsorted(a, grid, side="right") / len(a) - np.searchsorted(b, grid, side="right") / len(b)).max())
        crit = 1.95 * np.sqrt((len(a) + len(b)) / (len(a) * len(b)))
        print(f"   {label:78s} {tot[succ].mean():9.1f} {tot[~succ].mean():9.1f} {tot.mean():9.1f} {ks:12.3f} (critical {crit:.3f}){'  <- consistent' if ks < crit else ''}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=3000)
    args = ap.parse_args()
    report("obs", _abi.ENV_OBS, args.trials, 2)
    report("sta", _abi.ENV_STA, args.trials, 5)
