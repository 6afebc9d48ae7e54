#!/usr/bin/env python3
"""Diagnostic (GPU box): device EPA vs oracle EPA on overlapping pairs through urgym_probe_closest, then an Obs rollout
that prints the envs whose reward / link distances disagree.  Not part of the test suite."""
import os
import sys

import numpy as np
import torch
from scipy.spatial.transform import Rotation as Rot

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from ur_gym_amd import _abi, make_vec  # noqa: E402


def probes():
    env = make_vec("UR5DynReach-v1", num_envs=64, device="cuda:0", seed=1)
    rng = np.random.default_rng(7)
    H, C_, B_ = ob.HULL, ob.CYLZ, ob.BOX
    cases = []
    for i in range(600):
        pa = np.r_[rng.uniform(-0.5, 0.5, 3) + [0.5, 0, 0.35], Rot.random(random_state=int(rng.integers(1 << 30))).as_quat()]
        pb = np.r_[pa[:3] + rng.normal(0, 0.04, 3), Rot.random(random_state=int(rng.integers(1 << 30))).as_quat()]
        k = i % 4
        if k == 0: cases.append((H, [int(rng.integers(2, 7)), 0, 0], pa, C_, [0.05, 0.4, 0], pb))
        elif k == 1: cases.append((H, [int(rng.integers(2, 7)), 0, 0], pa, B_, [0.1, 0.55, 0.06], pb))
        elif k == 2: cases.append((H, [int(rng.integers(1, 4)), 0, 0], pa, H, [int(rng.integers(3, 7)), 0, 0], pb))
        else: cases.append((B_, [0.025, 0.025, 0.025], pa, C_, [0.05, 0.4, 0], pb))
    d, info = env.probe_closest([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], [c[3] for c in cases],
                                [c[4] for c in cases], [c[5] for c in cases], threshold=5.0)
    worst, npen, bad = 0.0, 0, 0
    for k, c in enumerate(cases):
        ref = ob.closest(c[0], c[1], c[2], c[3], c[4], c[5], threshold=5.0)
        if ref["penetrating"] != bool(info[k] & 1):
            print("penetration flag differs", k, ref, d[k], info[k])
            continue
        if ref["penetrating"]:
            npen += 1
            e = abs(d[k] - ref["distance"])
            worst = max(worst, e)
            if e > 1e-8:
                bad += 1
                if bad < 15:
                    print(f"case {k} kind {k % 4}: device {d[k]:.12f} oracle {ref['distance']:.12f} diff {e:.3e} info {info[k]} oracle-iters {ob.last_epa_iterations()}")
    print(f"probes: {npen} penetrating, worst |device - oracle| = {worst:.3e}, > 1e-8: {bad}")
    env.close()


def rollout(env_id, kind, n=320, steps=60, seed=23, **kw):
    env = make_vec(env_id, num_envs=n, device="cuda:0", seed=seed, **kw)
    orc = ob.OracleEnv(kind, n, threads=8, **kw)
    env.reset(seed=seed)
    orc.reset(seed=seed)
    rng = np.random.default_rng(seed)
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        env.step(torch.from_numpy(a).cuda())
        orc.step(a)
        torch.cuda.synchronize()
        r = env.buf["reward"].cpu().numpy()
        bad = np.nonzero(np.abs(r - orc.buf["reward"]) > 1e-4)[0]
        for i in bad[:6]:
            print(f"{env_id} step {t} env {i}: reward device {r[i]:.6f} oracle {orc.buf['reward'][i]:.6f} term {orc.buf['terminated'][i]} coll "
                  f"{orc.buf['collision'][i]} / {int(env.buf['collision'][i])} status {int(env.buf['status'][i])} / {orc.buf['status'][i]}")
        ld = env.buf["link_dist"].cpu().numpy()
        dl = np.abs(ld - orc.buf["link_dist"])
        if dl.max() > 1e-8:
            i = np.unravel_index(dl.argmax(), dl.shape)
            print(f"{env_id} step {t}: link_dist differs most at link {i[0] + 2} env {i[1]}: {ld[i]:.10f} vs {orc.buf['link_dist'][i]:.10f}")
        env.buf["link_dist"].copy_(torch.from_numpy(orc.buf["link_dist"]).cuda())
        for k in ("q", "obst_pos", "obst_quat", "goal", "obst_start", "obst_end", "obst_vel", "step_count", "episode_id"):
            env.buf[k].copy_(torch.from_numpy(orc.buf[k]).cuda())
    print(env_id, kw, "rollout done")
    env.close()


if __name__ == "__main__":
    probes()
    rollout("UR5ObsReach-v1", _abi.ENV_OBS, auto_reset=0)
    rollout("UR5ObsReach-v1", _abi.ENV_OBS)
    rollout("UR5DynReach-v1", _abi.ENV_DYN, check_collision=0, auto_reset=0)
