"""Shared helpers of the reference-pin tests (CPU: tests/test_reference_pins.py, GPU: tests/test_gpu_reference_pins.py).

tests/golden/reference_observations.json holds, per env, two CONSECUTIVE observations the reference's own PyBullet
environment produced (tests/golden/gen_reference_observations.py explains where they come from).  From the observation
BEFORE a step this module rebuilds the env state, so that one ``step()`` of the build can be compared with the observation
AFTER it.  Only the action is not recorded; it is recovered from the joint increment (q_after - q_before) / (0.1 pi), which
the float32 products of UR5Ori.set_action (UR5.py:273-279) reproduce to ~1e-7 rad.
"""
import json
import os

import numpy as np
from scipy.spatial.transform import Rotation as Rot

from ur_gym_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))

KIND = {"ori": _abi.ENV_ORI, "obs": _abi.ENV_OBS, "sta": _abi.ENV_STA, "dyn": _abi.ENV_DYN}
# which link_dist rule produced the stored observation (include/urgym.h URGYM_LINK_DIST_*): the Obs / Sta checkpoints date
# from Sep 2023, when get_link_distances still measured "workbench, obstacle and UR5" (its docstring, pyb_setup.py:440)
SCOPE = {"obs": _abi.LINK_DIST_WORKBENCH, "sta": _abi.LINK_DIST_WORKBENCH, "dyn": _abi.LINK_DIST_OBSTACLE}
# observation slots: (goal, obstacle pose xyz+rpy, velocity, link_dist)
SLOTS = {
    "ori": dict(goal=slice(12, 18)),
    "obs": dict(goal=slice(12, 15), obstacle=slice(15, 21), link_dist=slice(21, 26)),
    "sta": dict(goal=slice(12, 18), obstacle=slice(18, 24), link_dist=slice(24, 29)),
    "dyn": dict(goal=slice(12, 18), obstacle=slice(18, 24), velocity=slice(24, 30), link_dist=slice(30, 35)),
}


def load():
    with open(os.path.join(HERE, "golden", "reference_observations.json")) as f:
        raw = json.load(f)
    out = {}
    for name, e in raw.items():
        out[name] = {t: np.array(e[t]["observation"], dtype=np.float32) for t in ("before", "after")}
    return out


def bullet_quat(rpy):
    """pybullet getQuaternionFromEuler: R = Rz(yaw) Ry(pitch) Rx(roll) == scipy extrinsic 'xyz'."""
    return Rot.from_euler("xyz", np.asarray(rpy, dtype=np.float64)).as_quat()


def wrap(d):
    """difference of angles modulo 2 pi"""
    return (np.asarray(d, dtype=np.float64) + np.pi) % (2 * np.pi) - np.pi


def state_before(name, before, link_dist_state):
    """Arrays (SoA [field][1]) of the env state that produced observation `before`; `link_dist_state` = task.link_dist at that
    moment (evaluated by the caller at the same joint vector / obstacle pose: it is what observation `after` shows)."""
    o = before.astype(np.float64)
    sl = SLOTS[name]
    st = {"q": o[6:12].reshape(6, 1), "step_count": np.array([5], np.int32), "episode_id": np.array([1], np.int32)}
    goal = np.zeros(6)
    g = o[sl["goal"]]
    goal[: len(g)] = g
    st["goal"] = goal.reshape(6, 1)
    if name == "ori":
        return st
    pose = o[sl["obstacle"]]
    quat = bullet_quat(pose[3:])
    st["obst_pos"] = pose[:3].reshape(3, 1)
    st["obst_quat"] = quat.reshape(4, 1)
    st["link_dist"] = np.asarray(link_dist_state, dtype=np.float64).reshape(5, 1)
    start, end = pose.copy(), np.zeros(6)
    vel6 = np.zeros(6)  # the twist only: the library derives the displacement per env step from it (urgym_derive_obstacle_motion)
    if name == "dyn":
        # a start/end pair whose ReachDyn.set_velocity twist (reach.py:735-745, time_duration 2) is the stored one
        v = o[sl["velocity"]]
        end[:3] = start[:3] + 2.0 * v[:3]
        r_end = Rot.from_rotvec(2.0 * v[3:]) * Rot.from_quat(quat)
        end[3:] = r_end.as_euler("xyz")
        vel6[:] = v
    st["obst_start"] = start.reshape(6, 1)
    st["obst_end"] = end.reshape(6, 1)
    st["obst_vel"] = vel6.reshape(6, 1)
    return st


def action_between(before, after):
    return ((after[6:12].astype(np.float64) - before[6:12].astype(np.float64)) / (0.1 * np.pi)).astype(np.float32).reshape(1, 6)


def compare_after(name, got, after):
    """max abs deviation per block of the observation row (Euler angles modulo 2 pi)"""
    got = np.asarray(got, dtype=np.float64).ravel()
    ref = after.astype(np.float64)
    sl = SLOTS[name]
    dev = {"ee_pos": np.abs(got[0:3] - ref[0:3]).max(), "ee_rpy": np.abs(wrap(got[3:6] - ref[3:6])).max(),
           "q": np.abs(got[6:12] - ref[6:12]).max(), "goal": np.abs(got[sl["goal"]] - ref[sl["goal"]]).max()}
    if name != "ori":
        ob = sl["obstacle"]
        dev["obst_pos"] = np.abs(got[ob][:3] - ref[ob][:3]).max()
        dev["obst_rpy"] = np.abs(wrap(got[ob][3:] - ref[ob][3:])).max()
        dev["link_dist"] = np.abs(got[sl["link_dist"]] - ref[sl["link_dist"]]).max()
    if name == "dyn":
        dev["velocity"] = np.abs(got[sl["velocity"]] - ref[sl["velocity"]]).max()
    return dev
