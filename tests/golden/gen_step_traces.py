#!/usr/bin/env python3
"""Generate tests/golden/step_trace_{ori,obs,dyn,sta,obs_wb,sta_wb}.npz with the CPU oracle (seeded), as regression vectors
for the HIP path (*_wb: link_dist_scope = URGYM_LINK_DIST_WORKBENCH, the rule of the reference's Sep-2023 checkpoints).
These are outputs of the build's own oracle, not of the reference (which cannot run here: no pybullet; what the reference
itself pins is in tests/golden/reference_observations.json):

    python tests/golden/gen_step_traces.py

Each file: the post-reset state, K action batches, and per step observation / reward / flags, plus the final state.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from ur_gym_amd import _abi  # noqa: E402

N, K, SEED = 48, 36, 20241004
STATE = ("q", "goal", "obst_start", "obst_end", "obst_pos", "obst_quat", "obst_vel", "link_dist", "step_count", "episode_id")


def main():
    for name, kind, scope in (("ori", _abi.ENV_ORI, 0), ("obs", _abi.ENV_OBS, 0), ("dyn", _abi.ENV_DYN, 0), ("sta", _abi.ENV_STA, 0),
                              ("obs_wb", _abi.ENV_OBS, 1), ("sta_wb", _abi.ENV_STA, 1)):
        env = ob.OracleEnv(kind, N, threads=4, link_dist_scope=scope)
        env.reset(seed=SEED)
        if kind == _abi.ENV_STA:  # make a third of the obstacles move (the 18-column form of set_goal_and_obstacle, reach.py:492-503)
            mv = np.arange(0, N, 3)
            env.buf["obst_end"][:, mv] = env.buf["obst_start"][:, mv] + np.array([[0.15], [0.25], [0.1], [0.4], [-0.3], [0.0]])
            env.refresh()
        rng = np.random.default_rng(SEED + kind)
        out = {"seed": SEED, "kind": kind, "link_dist_scope": scope}
        for k in STATE:
            out["reset_" + k] = env.buf[k].copy()
        out["reset_observation"] = env.buf["observation"].copy()
        acts = (rng.uniform(-1, 1, (K, N, 6)) * rng.choice([0.3, 1.0, 1.7], (K, N, 1))).astype(np.float32)  # some beyond +-1: clipped
        rec = {k: [] for k in ("observation", "achieved_goal", "desired_goal", "reward", "terminated", "truncated", "is_success",
                               "collision", "final_observation", "link_dist", "q", "obst_pos", "obst_quat")}
        for t in range(K):
            env.step(acts[t])
            for k in rec:
                rec[k].append(env.buf[k].copy())
        out["actions"] = acts
        for k, v in rec.items():
            out["step_" + k] = np.stack(v)
        for k in STATE:
            out["final_" + k] = env.buf[k].copy()
        out["final_status"] = env.buf["status"].copy()
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"step_trace_{name}.npz")
        np.savez_compressed(path, **out)
        done = out["step_terminated"].sum() + out["step_truncated"].sum()
        print(name, "episodes finished in trace:", int(done), "size", os.path.getsize(path))
        env.close()


if __name__ == "__main__":
    main()
