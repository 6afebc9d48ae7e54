"""Diagnostic (CPU, oracle): GJK iteration counts of the exact link <-> obstacle queries of a UR5DynReach-v1 rollout, per link.
Answers: which links carry the long queries (static ticket order), and how the counts are distributed.
    python tools/diag/iter_census.py [N] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import binding as ob
from ur_gym_amd import _abi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
env = ob.OracleEnv(_abi.ENV_DYN, n, threads=8)
env.reset(seed=3)
rng = np.random.default_rng(3)
its = [[] for _ in range(5)]
def mat_to_quat(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(R).as_quat()
for s in range(steps):
    env.step(rng.uniform(-1, 1, (n, 6)).astype(np.float32))
    if s % 3 != 2: continue
    q = env.buf["q"]; op = env.buf["obst_pos"]; oq = env.buf["obst_quat"]
    for e in range(n):
        R, t = ob.fk(q[:, e])
        opose = np.concatenate([op[:, e], oq[:, e]])
        for li in range(5):
            link = 2 + li
            pose = np.concatenate([t[link], mat_to_quat(R[link])])
            r = ob.closest(ob.HULL, [link], pose, ob.CYLZ, [0.05, 0.4], opose)
            its[li].append(r["iterations"])
for li in range(5):
    a = np.array(its[li])
    print(f"link {2 + li}: n {len(a)} mean {a.mean():.2f} p50 {np.median(a):.0f} p90 {np.percentile(a, 90):.0f} p99 {np.percentile(a, 99):.0f} max {a.max()}  share >= 16: {100 * (a >= 16).mean():.2f} %  >= 24: {100 * (a >= 24).mean():.3f} %")
allv = np.concatenate([np.array(x) for x in its])
print(f"all: mean {allv.mean():.2f} p99 {np.percentile(allv, 99):.0f} max {allv.max()}")
