#!/usr/bin/env python3
"""Optional cross-check of the oracle against a real PyBullet.

The oracle of this repository restates what the reference obtains from pybullet (Bullet3) — forward kinematics, Euler /
quaternion conventions, `getClosestPoints` distances between convex hulls, cylinders and boxes with Bullet's margins, the
20-substep motion of the mass-0 obstacle — but no pybullet exists on the machines this was built on, so that boundary is
"parity unpinned" (DESIGN.md §3).  On a machine that HAS pybullet this driver measures it item by item:

    python tools/pybullet_crosscheck.py --reference /path/to/UR-gym  [--samples 2000] [--seed 0]

It is build-authored: it imports nothing from the reference's Python; it only lets pybullet load the reference's URDF and
collision meshes (`UR_gym/envs/robots/urdf/ur5e.urdf`) and creates the scene bodies with the numbers quoted from
reach.py / pyb_setup.py below.  `--self-test` exercises everything except the pybullet calls (what CI can run here).

Exit code 0 when every item is within its tolerance, 1 otherwise, 2 when pybullet is missing.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# scene constants (reach.py:266-305, 613-651; pyb_setup.py:780-845): box half extents / centres, obstacle cylinder
TABLE = dict(half=[0.55, 0.9, 0.46], pos=[0.5, 0.0, -0.12 - 0.46])
TRACK = dict(half=[0.1, 0.55, 0.06], pos=[0.0, 0.0, 0.0 - 0.06])
OBSTACLE = dict(radius=0.05, height=0.4)
NEUTRAL = np.array([0.0, -1.5708, 0.0, -1.5708, 0.0, 0.0])  # UR5.py:262
TOL = {"fk_position": 1e-9, "fk_rotation": 1e-9, "euler": 1e-9, "quaternion": 1e-12, "link_distance": 1e-4,
       "collision_verdicts": 0, "target_clearance": 1e-4, "obstacle_motion": 1e-9}


def sample_cases(rng, n):
    """Joint vectors around the workspace the tasks visit + obstacle poses in the Dyn ranges (reach.py:584-587)."""
    q = NEUTRAL + rng.uniform(-1.5, 1.5, (n, 6))
    pos = rng.uniform([0.5, -0.8, 0.25], [1.2, 0.8, 0.75], (n, 3))
    roll = np.where(rng.random(n) < 0.5, rng.uniform(-150, -30, n), rng.uniform(30, 150, n))
    pitch = np.where(np.abs(roll) > 90, rng.uniform(-150, -30, n), rng.uniform(30, 150, n))
    rpy = np.deg2rad(np.stack([roll, pitch, np.zeros(n)], axis=1))
    return q, pos, rpy


def oracle_side(q, pos, rpy):
    from oracle import binding as ob

    out = {"fk": [], "euler": [], "ld": [], "coll": [], "quat": []}
    for i in range(len(q)):
        R, t = ob.fk(q[i])
        quat = ob.quat_from_euler(rpy[i])
        ld, coll, _ = ob.query(q[i], np.r_[pos[i], quat])
        out["fk"].append((R, t))
        out["quat"].append(quat)
        out["euler"].append(ob.euler_from_quat(quat))
        out["ld"].append(ld)
        out["coll"].append(coll)
    return out


def pybullet_side(reference, q, pos, rpy):
    import pybullet as p

    cid = p.connect(p.DIRECT)
    urdf = os.path.join(reference, "UR_gym", "envs", "robots", "urdf", "ur5e.urdf")
    robot = p.loadURDF(urdf, basePosition=[0, 0, 0], useFixedBase=True, physicsClientId=cid)

    def box(spec):
        col = p.createCollisionShape(p.GEOM_BOX, halfExtents=spec["half"], physicsClientId=cid)
        return p.createMultiBody(baseMass=0.0, baseCollisionShapeIndex=col, basePosition=spec["pos"], physicsClientId=cid)

    table, track = box(TABLE), box(TRACK)
    col = p.createCollisionShape(p.GEOM_CYLINDER, radius=OBSTACLE["radius"], height=OBSTACLE["height"], physicsClientId=cid)
    obstacle = p.createMultiBody(baseMass=0.0, baseCollisionShapeIndex=col, basePosition=[0, 0, 1.0], physicsClientId=cid)
    out = {"fk": [], "euler": [], "ld": [], "coll": [], "quat": []}
    for i in range(len(q)):
        for j in range(6):
            p.resetJointState(robot, j + 1, float(q[i, j]), physicsClientId=cid)
        quat = p.getQuaternionFromEuler([float(x) for x in rpy[i]])
        p.resetBasePositionAndOrientation(obstacle, [float(x) for x in pos[i]], quat, physicsClientId=cid)
        links = []
        for link in range(1, 8):
            st = p.getLinkState(robot, link, computeForwardKinematics=True, physicsClientId=cid)
            links.append((np.array(p.getMatrixFromQuaternion(st[5])).reshape(3, 3), np.array(st[4])))
        out["fk"].append(links)
        out["quat"].append(np.array(quat))
        out["euler"].append(np.array(p.getEulerFromQuaternion(quat)))
        out["ld"].append(np.array([p.getClosestPoints(robot, obstacle, distance=5.0, linkIndexA=l, physicsClientId=cid)[0][8] for l in range(2, 7)]))
        hit = any(len(p.getClosestPoints(robot, obstacle, distance=0.01, linkIndexA=l, physicsClientId=cid)) > 0 for l in range(2, 7))
        for body in (table, track):
            hit = hit or any(len(p.getClosestPoints(robot, body, distance=0.01, linkIndexA=l, physicsClientId=cid)) > 0 for l in range(2, 7))
        for la, lbs in ((1, (3, 4, 5, 6)), (2, (4, 5, 6)), (3, (5, 6))):
            hit = hit or any(len(p.getClosestPoints(robot, robot, distance=0.01, linkIndexA=la, linkIndexB=lb, physicsClientId=cid)) > 0 for lb in lbs)
        out["coll"].append(bool(hit))
    # obstacle motion: pyb_setup.py:25,39-41,52-55 — 20 substeps of 1/500 s with a base velocity on a mass-0 body
    p.setTimeStep(1.0 / 500.0, physicsClientId=cid)
    motion = []
    for i in range(min(len(q), 64)):
        quat = p.getQuaternionFromEuler([float(x) for x in rpy[i]])
        p.resetBasePositionAndOrientation(obstacle, [float(x) for x in pos[i]], quat, physicsClientId=cid)
        vel = _test_velocity(i)
        p.resetBaseVelocity(obstacle, linearVelocity=list(vel[:3]), angularVelocity=list(vel[3:]), physicsClientId=cid)
        for _ in range(20):
            p.stepSimulation(physicsClientId=cid)
        pp, qq = p.getBasePositionAndOrientation(obstacle, physicsClientId=cid)
        motion.append(np.r_[pp, qq])
    out["motion"] = np.array(motion)
    p.disconnect(cid)
    return out


def _test_velocity(i):
    rng = np.random.default_rng(1000 + i)
    return np.r_[rng.uniform(-0.5, 0.5, 3), rng.uniform(-1.0, 1.0, 3)]


def oracle_motion(pos, rpy, count):
    """The oracle's obstacle update for one env step (20 sub-steps incl. the omega x v drift of the base's linear velocity)."""
    from oracle import binding as ob

    return np.array([ob.integrate_obstacle(np.r_[pos[i], ob.quat_from_euler(rpy[i])], _test_velocity(i)) for i in range(count)])


def compare(a, b):
    rows = []
    n = len(a["ld"])
    dpos = max(np.abs(a["fk"][i][1][l] - b["fk"][i][l - 1][1]).max() for i in range(n) for l in range(1, 7))
    drot = max(np.abs(a["fk"][i][0][l] - b["fk"][i][l - 1][0]).max() for i in range(n) for l in range(1, 7))
    rows.append(("fk_position", dpos))
    rows.append(("fk_rotation", drot))
    rows.append(("quaternion", max(min(np.abs(a["quat"][i] - b["quat"][i]).max(), np.abs(a["quat"][i] + b["quat"][i]).max()) for i in range(n))))
    de = np.abs(np.array(a["euler"]) - np.array(b["euler"]))
    rows.append(("euler", float(np.minimum(de, np.abs(de - 2 * np.pi)).max())))
    rows.append(("link_distance", float(np.abs(np.array(a["ld"]) - np.array(b["ld"])).max())))
    rows.append(("collision_verdicts", int(np.sum(np.array(a["coll"]) != np.array(b["coll"])))))
    if "motion" in a and "motion" in b:
        dm = np.abs(a["motion"] - b["motion"])
        dm[:, 3:] = np.minimum(dm[:, 3:], np.abs(a["motion"][:, 3:] + b["motion"][:, 3:]))
        rows.append(("obstacle_motion", float(dm.max())))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", help="path of a WanqingXia/UR-gym checkout (for its URDF + meshes)")
    ap.add_argument("--samples", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--self-test", action="store_true", help="run the oracle side only and compare it with itself")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    q, pos, rpy = sample_cases(rng, args.samples)
    mine = oracle_side(q, pos, rpy)
    mine["motion"] = oracle_motion(pos, rpy, min(args.samples, 64))
    if args.self_test:
        other = {k: v for k, v in mine.items()}
        other["fk"] = [[(R[min(l, 6)], t[min(l, 6)]) for l in range(1, 8)] for (R, t) in mine["fk"]]  # ee_link (7) == link 6 frame
    else:
        try:
            import pybullet  # noqa: F401
        except ImportError:
            print("pybullet is not installed: nothing to cross-check against (parity stays unpinned).")
            return 2
        if not args.reference:
            ap.error("--reference is required")
        other = pybullet_side(args.reference, q, pos, rpy)
    ok = True
    print(f"{'item':22s} {'max |oracle - pybullet|':>26s} {'tolerance':>10s}")
    for name, val in compare(mine, other):
        good = val <= TOL[name]
        ok = ok and good
        print(f"{name:22s} {val:26.3e} {TOL[name]:10.1e}  {'ok' if good else 'DIFFERS'}")
    print("collisions among the samples:", int(np.sum(mine["coll"])), "of", len(q))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
