#!/usr/bin/env python3
"""Diagnostic (GPU box): time of one wave-cooperative EPA, from urgym_probe_closest on all-penetrating batches."""
import os, sys, time
import numpy as np, torch
from scipy.spatial.transform import Rotation as Rot
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ur_gym_amd import make_vec

env = make_vec("UR5DynReach-v1", num_envs=64, device="cuda:0", seed=1)
rng = np.random.default_rng(0)
def batch(n, overlap):
    pa = np.c_[rng.uniform(-0.3, 0.3, (n, 3)) + [0.5, 0, 0.4], Rot.random(n, random_state=1).as_quat()]
    off = rng.normal(0, 0.02, (n, 3)) if overlap else rng.normal(0, 0.02, (n, 3)) + [0.6, 0.0, 0.0]
    pb = np.c_[pa[:, :3] + off, Rot.random(n, random_state=2).as_quat()]
    links = rng.integers(2, 7, n)
    return (np.zeros(n, int), np.c_[links, np.zeros((n, 2))], pa, np.ones(n, int), np.tile([0.05, 0.4, 0.0], (n, 1)), pb)
for overlap in (False, True):
    n = 64 * 64
    args = batch(n, overlap)
    d, info = env.probe_closest(*args)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d, info = env.probe_closest(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    npen = int((info & 1).sum())
    print(f"overlap={overlap}: {n} probes in 64-lane workgroups, {npen} penetrating, {dt * 1e3:.3f} ms per call -> "
          f"{(dt * 1e6) / max(1, npen / 64):.1f} us per EPA if serial within a workgroup (64 workgroups run side by side)")
env.close()
