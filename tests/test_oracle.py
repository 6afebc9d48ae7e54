"""CPU tests (-m "not gpu"): the oracle against the reference's golden vectors and independent evaluations.

What pins the oracle:
  * utils.distance / utils.angular_distance: fixtures produced by the reference's own UR_gym/utils.py
    (tests/golden/utils_golden.json, generator committed next to it)                                   -> exact
  * forward kinematics: an independent scipy evaluation of the URDF chain + the survey's scratch values -> 1e-12
  * Bullet quaternion/Euler conventions: scipy Rotation                                                 -> 1e-12
  * closest distances (GJK restatement): analytic cases and an independent constrained optimiser         -> 1e-6
  * task semantics (lagged link_dist, early returns, obstacle motion, truncation, reset rules): by construction
  * at the pybullet boundary itself: tests/test_reference_pins.py (observations of the reference's own PyBullet envs)
What no reference value exists for is listed in DESIGN.md section 3.
"""
import json
import os

import numpy as np
import pytest
from scipy.optimize import minimize
from scipy.spatial.transform import Rotation as Rot

from ur_gym_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MODEL = np.load(os.path.join(ROOT, "data", "ur5e_model.npz"))
NEUTRAL = np.array([0.0, -1.5708, 0.0, -1.5708, 0.0, 0.0])


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "utils_golden.json")) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------------------ a7: utils.py
def test_distance_matches_reference_utils(oracle, golden):
    a, b = np.array(golden["a"]), np.array(golden["b"])
    got = np.array([oracle.distance(x, y) for x, y in zip(a, b)])
    assert np.max(np.abs(got - np.array(golden["distance_single"]))) < 1e-14
    assert np.max(np.abs(got - np.array(golden["distance_batch"]))) < 1e-14
    got3 = np.array([oracle.distance(np.r_[x[:3], 0, 0, 0], np.r_[y[:3], 0, 0, 0]) for x, y in zip(a[:20], b[:20])])
    assert np.max(np.abs(got3 - np.array(golden["distance_3vec"]))) < 1e-14


def test_angular_distance_matches_reference_utils(oracle, golden):
    a, b = np.array(golden["a"]), np.array(golden["b"])
    got = np.array([oracle.angular_distance(x, y) for x, y in zip(a, b)])
    ref = np.array(golden["angular_single"])
    # 2*acos(|dot|) is ill-conditioned at theta -> 0: both sides evaluate the same formula in float64, so the
    # agreement is a few ulp of the DOT product, i.e. ~1e-8 rad at theta ~ 0 and 1e-15 elsewhere
    assert np.max(np.abs(got - ref)) < 5e-8
    well = ref > 1e-3
    assert np.max(np.abs(got[well] - ref[well])) < 1e-12
    assert np.max(np.abs(got - np.array(golden["angular_batch"]))) < 5e-8


def test_survey_scratch_values(oracle):
    # SURVEY.md §8c "What IS importable here": values obtained from the reference utils.py
    assert abs(oracle.distance([.1, .2, .3, 0, 0, 0], [.4, 0, .1, 0, 0, 0]) - 0.41231056) < 1e-8
    assert abs(oracle.angular_distance([0, 0, 0, -2, 0, -1], [0, 0, 0, -3, 0, -0.2]) - 1.25905066) < 1e-8


def test_sampler_ranges_match_reference(oracle, golden):
    """Reset samplers (utils.py:81-100) reproduce the reference's ranges and its roll/pitch sign rule."""
    env = oracle.OracleEnv(_abi.ENV_DYN, 4096)
    env.reset(seed=7)
    goal, start = env.buf["goal"], env.buf["obst_start"]
    con, obs = golden["sampler_constrained"], golden["sampler_obstacle"]
    # goal rpy = deg2rad(U(-180,-90), 0, U(-180,0))
    assert goal[3].min() >= -np.pi and goal[3].max() <= -np.pi / 2 and np.all(goal[4] == 0)
    assert goal[5].min() >= -np.pi and goal[5].max() <= 0
    assert abs(goal[3].min() - con["min"][0]) < 0.02 and abs(goal[3].max() - con["max"][0]) < 0.02
    assert abs(goal[5].min() - con["min"][2]) < 0.02 and abs(goal[5].max() - con["max"][2]) < 0.02
    # obstacle: |roll| in [30,150] deg, pitch negative iff |roll| > 90 deg, yaw 0
    assert np.all(np.abs(start[3]) >= np.deg2rad(30) - 1e-12) and np.all(np.abs(start[3]) <= np.deg2rad(150) + 1e-12)
    assert np.all((np.abs(start[3]) > np.pi / 2) == (start[4] < 0)) and obs["pitch_sign_rule_ok"]
    assert np.all(start[5] == 0)
    assert abs((start[3] < 0).mean() - 0.5) < 0.05 and abs(obs["frac_negative_roll"] - 0.5) < 0.05
    env.close()


# ------------------------------------------------------------------------------------------------ a5: FK
def fk_scipy(q):
    T = np.eye(4)
    out = []
    for k in range(6):
        A = np.eye(4)
        A[:3, :3] = Rot.from_euler("xyz", MODEL["joint_rpy"][k]).as_matrix()  # URDF rpy = extrinsic xyz
        A[:3, 3] = MODEL["joint_xyz"][k]
        B = np.eye(4)
        B[:3, :3] = Rot.from_euler("z", q[k]).as_matrix()
        T = T @ A @ B
        out.append(T.copy())
    return out


def test_fk_against_independent_scipy_chain(oracle):
    rng = np.random.default_rng(0)
    for _ in range(100):
        q = rng.uniform(-2 * np.pi, 2 * np.pi, 6)
        R, t = oracle.fk(q)
        ref = fk_scipy(q)
        for k in range(6):
            assert np.max(np.abs(R[k + 1] - ref[k][:3, :3])) < 1e-12
            assert np.max(np.abs(t[k + 1] - ref[k][:3, 3])) < 1e-12


def test_fk_survey_pins(oracle):
    # neutral pose and q=0 end-effector values from a scratch scipy evaluation of ur5e.urdf
    ee = oracle.ee_pose(NEUTRAL)
    assert np.max(np.abs(ee[:3] - [-0.000673, -0.232972, 1.080159])) < 1e-6
    R, _ = oracle.fk(NEUTRAL)
    q = Rot.from_matrix(R[6]).as_quat()
    ref = np.array([0.001181, 0.707196, -0.707016, -0.001176])
    assert min(np.max(np.abs(q - ref)), np.max(np.abs(q + ref))) < 1e-6
    assert np.max(np.abs(oracle.ee_pose(np.zeros(6))[:3] - [-0.817267, -0.234444, 0.062675])) < 1e-6


# ------------------------------------------------------------------------------------------------ Bullet conventions
def test_quaternion_from_euler_is_zyx(oracle):
    rng = np.random.default_rng(1)
    for _ in range(100):
        rpy = rng.uniform(-np.pi, np.pi, 3)
        q = oracle.quat_from_euler(rpy)
        ref = Rot.from_euler("xyz", rpy).as_quat()  # extrinsic xyz == Rz(y) Ry(p) Rx(r)
        assert min(np.max(np.abs(q - ref)), np.max(np.abs(q + ref))) < 1e-14


def test_euler_from_quaternion_round_trip_and_gimbal_branch(oracle):
    rng = np.random.default_rng(2)
    for _ in range(200):
        rpy = np.array([rng.uniform(-np.pi, np.pi), rng.uniform(-1.5, 1.5), rng.uniform(-np.pi, np.pi)])
        back = oracle.euler_from_quat(oracle.quat_from_euler(rpy))
        assert np.max(np.abs(back - rpy)) < 1e-9
    # beyond +-90 deg pitch the read-back is the canonical equivalent triple (same rotation)
    for _ in range(50):
        rpy = rng.uniform(-np.pi, np.pi, 3)
        back = oracle.euler_from_quat(oracle.quat_from_euler(rpy))
        assert abs(back[1]) <= np.pi / 2 + 1e-12
        assert np.max(np.abs(Rot.from_euler("xyz", back).as_matrix() - Rot.from_euler("xyz", rpy).as_matrix())) < 1e-9
    # gimbal branch: |sin(pitch)| >= 0.99999 -> roll = 0, pitch = +-pi/2, yaw carries everything
    for sgn in (1.0, -1.0):
        rpy = np.array([0.3, sgn * (np.pi / 2 - 1e-4), -0.7])
        back = oracle.euler_from_quat(oracle.quat_from_euler(rpy))
        assert back[0] == 0.0 and abs(back[1] - sgn * np.pi / 2) < 1e-15


def test_dyn_velocity_is_half_the_start_to_end_twist(oracle):
    rng = np.random.default_rng(3)
    for _ in range(50):
        s = np.r_[rng.uniform(-1, 1, 3), rng.uniform(-2.6, 2.6, 2), 0.0]
        e = np.r_[rng.uniform(-1, 1, 3), rng.uniform(-2.6, 2.6, 2), 0.0]
        v = oracle.dyn_velocity(s, e, 2.0)
        assert np.max(np.abs(v[:3] - (e[:3] - s[:3]) / 2)) < 1e-15
        rel = Rot.from_euler("xyz", e[3:]) * Rot.from_euler("xyz", s[3:]).inv()  # world-frame delta
        assert np.max(np.abs(v[3:] - rel.as_rotvec() / 2)) < 1e-9


# ------------------------------------------------------------------------------------------------ a8/a9: distances
IDENT = [0, 0, 0, 1]


def hull_points(link):
    off = MODEL["hull_offset"]
    return MODEL["hull_verts"][off[link - 1]:off[link]]


def dual_distance_hull_vs_cylinder(link, R, t, core_r, core_h):
    """Independent check through convex duality: for separated convex sets
        dist(A, B) = max over unit n of [ min_{x in A} n.x  -  max_{y in B} n.y ],
    and every n gives a valid LOWER bound.  Only support functions are needed (brute force over the hull vertices,
    closed form for the cylinder); the GJK result is an UPPER bound (its closest point lies in A - B), so agreement
    of the two pins the distance from both sides.  (A primal constrained optimiser, scipy trust-constr on the hull's
    H-representation, agreed with the oracle to 1e-8 in a one-off check but takes a minute per case.)"""
    pts = hull_points(link) @ R.T + t

    def lower_bound(u):
        n = u / np.linalg.norm(u)
        return (pts @ n).min() - (core_r * np.hypot(n[0], n[1]) + core_h * abs(n[2]))

    best = -np.inf
    c = pts.mean(0)
    starts = [c, c - [0, 0, np.clip(c[2], -core_h, core_h)], pts[np.argmin(np.linalg.norm(pts, axis=1))]]
    for u0 in starts:
        res = minimize(lambda u: -lower_bound(u), u0 / np.linalg.norm(u0), method="Nelder-Mead",
                       options={"xatol": 1e-13, "fatol": 1e-15, "maxiter": 20000, "maxfev": 20000})
        res = minimize(lambda u: -lower_bound(u), res.x, method="Powell", options={"xtol": 1e-13, "ftol": 1e-15})
        best = max(best, -res.fun)
    return best


def test_sphere_cylinder_analytic(oracle):
    # sphere r=0.02 (point core) against the obstacle cylinder (core r=.049,h=.199, margin .001 = Bullet's default
    # collision margin, pinned by tests/test_reference_pins.py): the rim is rounded with a 1 mm radius
    cyl = [0.05, 0.4]
    for p, expect in [((0.3, 0, 0), 0.3 - 0.05 - 0.02), ((0, 0, 0.5), 0.5 - 0.2 - 0.02), ((0, -0.25, 0.1), 0.25 - 0.05 - 0.02)]:
        r = oracle.closest(oracle.SPHERE, [0.02], [*p, *IDENT], oracle.CYLZ, cyl, [0, 0, 0, *IDENT])
        assert abs(r["distance"] - expect) < 1e-9
    # diagonal off the rim: distance to the rim circle of the CORE minus both margins (rounded edge, App. A.5.6)
    p = np.array([0.2, 0.0, 0.4])
    core = np.hypot(p[0] - 0.049, p[2] - 0.199)
    r = oracle.closest(oracle.SPHERE, [0.02], [*p, *IDENT], oracle.CYLZ, cyl, [0, 0, 0, *IDENT])
    assert abs(r["distance"] - (core - 0.001 - 0.02)) < 1e-9


def test_box_box_like_cases_via_hull_track(oracle):
    # box (Dyn target, half .025 -> core .024 + margin .001) vs cylinder, axis-aligned face-to-face and rotated
    r = oracle.closest(oracle.BOX, [0.025] * 3, [0.3, 0, 0, *IDENT], oracle.CYLZ, [0.05, 0.4], [0, 0, 0, *IDENT])
    assert abs(r["distance"] - (0.3 - 0.025 - 0.05)) < 5e-9  # (Bullet's GJK stops at a relative 1e-12 on the SQUARED distance)
    q45 = Rot.from_euler("z", 45, degrees=True).as_quat()
    r = oracle.closest(oracle.BOX, [0.025] * 3, [0.3, 0, 0, *q45], oracle.CYLZ, [0.05, 0.4], [0, 0, 0, *IDENT])
    # the vertical box edge (rounded with radius .001) points at the cylinder
    assert abs(r["distance"] - (0.3 - (0.024 * np.sqrt(2) + 0.001) - 0.05)) < 1e-9


@pytest.mark.parametrize("link", [2, 3, 4, 5, 6])
def test_hull_cylinder_against_independent_optimiser(oracle, link):
    rng = np.random.default_rng(10 + link)
    for _ in range(4):
        Rm = Rot.random(random_state=int(rng.integers(1 << 30)))
        t = rng.uniform(-0.2, 0.2, 3) + np.array([0.45, 0.1, 0.0])
        pose = np.r_[t, Rm.as_quat()]
        got = oracle.closest(oracle.HULL, [link], pose, oracle.CYLZ, [0.05, 0.4], [0, 0, 0, *IDENT])
        lower = dual_distance_hull_vs_cylinder(link, Rm.as_matrix(), t, 0.049, 0.199) - 0.001 - 0.001
        assert not got["penetrating"]
        assert lower <= got["distance"] + 1e-9, (link, got, lower)   # duality: never above the true distance
        assert got["distance"] - lower < 2e-6, (link, got, lower)    # ... and the search closes the gap


def test_distance_is_rigid_motion_invariant_and_symmetric(oracle):
    rng = np.random.default_rng(4)
    for _ in range(20):
        pa = np.r_[rng.uniform(-0.3, 0.3, 3) + [0.5, 0, 0.3], Rot.random(random_state=int(rng.integers(1 << 30))).as_quat()]
        pb = np.r_[rng.uniform(-0.2, 0.2, 3), Rot.random(random_state=int(rng.integers(1 << 30))).as_quat()]
        d0 = oracle.closest(oracle.HULL, [3], pa, oracle.CYLZ, [0.05, 0.4], pb)["distance"]
        G = Rot.random(random_state=int(rng.integers(1 << 30)))
        g = rng.uniform(-1, 1, 3)
        pa2 = np.r_[G.apply(pa[:3]) + g, (G * Rot.from_quat(pa[3:])).as_quat()]
        pb2 = np.r_[G.apply(pb[:3]) + g, (G * Rot.from_quat(pb[3:])).as_quat()]
        d1 = oracle.closest(oracle.HULL, [3], pa2, oracle.CYLZ, [0.05, 0.4], pb2)["distance"]
        d2 = oracle.closest(oracle.HULL, [3], pa, oracle.HULL, [5], pb)["distance"]
        d3 = oracle.closest(oracle.HULL, [5], pb, oracle.HULL, [3], pa)["distance"]
        assert abs(d0 - d1) < 2e-6 and abs(d2 - d3) < 2e-6


def test_neutral_pose_is_collision_free(oracle):
    """reach.py:682 'Collision after reset, this should not happen': table/track/self pairs are clear at the neutral pose."""
    ld, coll, _ = oracle.query(NEUTRAL, None)
    assert not coll


def test_collision_rules(oracle):
    # elbow folded back onto the shoulder -> self collision; arm pushed into the table -> collision
    _, coll, _ = oracle.query([0, -1.5708, 3.0, -1.5708, 0, 0], None)
    assert coll
    _, coll, _ = oracle.query([np.pi, 0.3, 0.0, 0, 0, 0], None)  # arm swung over the table and lowered into it
    assert coll
    _, coll, _ = oracle.query([0, 0.4, 0.0, 0, 0, 0], None)  # same lift but away from the table (x < 0): free
    assert not coll
    # obstacle wrapped around the forearm -> collision through the obstacle rule only (Ori ignores it, pyb_setup.py:398)
    R, t = oracle.fk(NEUTRAL)
    mid = t[3] + R[3] @ np.array([-0.2, 0, 0.007])
    pose = np.r_[mid, 0, 0, 0, 1]
    ld, coll, status = oracle.query(NEUTRAL, pose)
    assert coll and ld[1] < 0.01


# ------------------------------------------------------------------------------------------------ task semantics
def test_philox_known_answers(oracle):
    """Philox4x32-10 known-answer vectors of Random123 (kat_vectors): counter/key all zero and all ones."""
    def raw(seed, env, ep, att):
        u = oracle.philox(seed, env, ep, att)
        return [int(round(x * 4294967296.0 - 0.5)) for x in u]

    z = raw(0, 0, 0, 0)  # block 0 = counter (0,0,0,0), key (0,0)
    assert z[:4] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    o = raw(0xFFFFFFFFFFFFFFFE, 0, 0, 0)  # different key changes everything
    assert o[:4] != z[:4]


def make(oracle, kind, n=64, seed=5, **kw):
    env = oracle.OracleEnv(kind, n, **kw)
    env.reset(seed=seed)
    return env


def test_observation_layout_and_reset(oracle):
    for kind, od, gd in ((0, 18, 6), (1, 26, 3), (2, 35, 6)):
        env = make(oracle, kind)
        obs = env.buf["observation"]
        assert obs.shape == (64, od) and env.buf["achieved_goal"].shape == (64, gd)
        # robot part: ee pose at the neutral configuration, then q (UR5.py:320-325)
        ee = oracle.ee_pose(NEUTRAL)
        assert np.max(np.abs(obs[:, :6] - ee.astype(np.float32))) < 1e-6
        assert np.max(np.abs(obs[:, 6:12] - NEUTRAL.astype(np.float32))) == 0
        assert np.all(env.buf["step_count"] == 0) and np.all(env.buf["episode_id"] == 1)
        assert np.all(env.buf["achieved_goal"] == obs[:, :gd])
        goal = env.buf["goal"][:gd].T.astype(np.float32)
        assert np.all(env.buf["desired_goal"] == goal)
        assert np.all(obs[:, 12:12 + gd] == goal)
        if kind == 2:
            # obstacle read-back pose, stale velocity (zeros on the first reset), fresh link distances
            assert np.all(obs[:, 18:21] == env.buf["obst_start"][:3].T.astype(np.float32))
            assert np.all(obs[:, 24:30] == 0)
            assert np.all(obs[:, 30:35] == env.buf["link_dist"].T.astype(np.float32))
            # rejection rules (reach.py:668-675)
            travel = np.linalg.norm(env.buf["obst_end"][:3] - env.buf["obst_start"][:3], axis=0)
            assert np.all(travel >= 1.0)
        env.close()


def test_link_dist_lags_one_step_and_dyn_motion(oracle):
    env = make(oracle, _abi.ENV_DYN, n=32, seed=9, auto_reset=0)
    ld_reset = env.buf["link_dist"].copy()
    start, end, vel = env.buf["obst_start"].copy(), env.buf["obst_end"].copy(), env.buf["obst_vel"].copy()
    a = np.zeros((32, 6), np.float32)
    env.step(a)
    # the step's observation still shows the reset distances (core.py:311 vs 316) ...
    assert np.all(env.buf["observation"][:, 30:35] == ld_reset.T.astype(np.float32))
    ok = (env.buf["terminated"] == 0)
    # ... while the state already holds the new ones where compute_reward ran to the end (reach.py:780-782)
    assert np.any(env.buf["link_dist"][:, ok] != ld_reset[:, ok])
    # velocity slot = the velocity applied in this step (reach.py:744-746)
    assert np.max(np.abs(env.buf["observation"][:, 24:30] - vel[:6].T.astype(np.float32))) == 0
    for _ in range(24):
        env.step(a)
    # after 25 steps the obstacle has moved 25 x the per-step displacement (rows 6..8 of obst_vel): ABOUT half of
    # start->end (reach.py:735-745, dt = 0.04) -- not exactly, because Bullet lets the base's linear velocity drift by
    # h * (omega x v) in each of the 20 sub-steps of an env step (pinned by tests/test_reference_pins.py)
    assert np.max(np.abs(env.buf["obst_pos"] - (start[:3] + 25.0 * vel[6:9]))) < 1e-12
    half = start[:3] + 0.5 * (end[:3] - start[:3])
    drift = np.abs(env.buf["obst_pos"] - half).max(axis=0)
    bound = 25 * 0.5 * 1.05 * 0.04 ** 2 * np.linalg.norm(np.cross(vel[3:6].T, vel[:3].T), axis=1)  # 25 steps x (21/40) dt^2 |w x v|
    assert np.all(drift <= bound + 1e-9) and np.any(drift > 1e-4)
    pos25 = env.buf["obst_pos"].copy()
    env.step(a)
    assert np.all(env.buf["observation"][:, 24:30] == 0)  # step 26: velocity zero (reach.py:748-752)
    assert np.all(env.buf["obst_pos"] == pos25)
    env.close()


def test_truncation_and_autoreset(oracle):
    env = make(oracle, _abi.ENV_ORI, n=16, seed=3)
    a = np.zeros((16, 6), np.float32)
    for k in range(99):
        env.step(a)
        assert not env.buf["truncated"].any()
    before = env.buf["observation"].copy()
    env.step(a)  # 100th step: TimeLimit (UR_gym/__init__.py:41)
    assert env.buf["truncated"].all() and not env.buf["terminated"].any()
    # auto-reset: terminal observation preserved, new episode started with a new goal
    assert np.all(env.buf["final_observation"][:, :12] == before[:, :12])
    assert np.all(env.buf["step_count"] == 0) and np.all(env.buf["episode_id"] == 2)
    assert np.any(env.buf["observation"][:, 12:18] != env.buf["final_observation"][:, 12:18])
    env.close()


def test_stale_velocity_survives_reset(oracle):
    """ReachDyn.reset() does not clear self.velocity (reach.py:664-683): the first observation of the next episode
    still carries the last step's velocity."""
    env = make(oracle, _abi.ENV_DYN, n=64, seed=21)
    rng = np.random.default_rng(0)
    seen = False
    for _ in range(40):
        env.step(rng.uniform(-1, 1, (64, 6)).astype(np.float32))
        done = (env.buf["terminated"] | env.buf["truncated"]).astype(bool)
        if done.any():
            fin = env.buf["final_observation"][done]
            new = env.buf["observation"][done]
            assert np.all(new[:, 24:30] == fin[:, 24:30])
            seen = seen or np.any(fin[:, 24:30] != 0)
    assert seen
    env.close()


def test_dyn_early_return_keeps_link_dist(oracle):
    """reach.py:766-770: on collision/success compute_reward returns before link_dist is refreshed."""
    env = make(oracle, _abi.ENV_DYN, n=256, seed=2, auto_reset=0)
    rng = np.random.default_rng(1)
    hit = False
    for _ in range(60):
        before = env.buf["link_dist"].copy()
        env.step(rng.uniform(-1, 1, (256, 6)).astype(np.float32))
        coll = env.buf["collision"].astype(bool)
        if coll.any():
            hit = True
            assert np.all(env.buf["link_dist"][:, coll] == before[:, coll])
            assert np.all(env.buf["reward"][coll] == -500.0)
    assert hit
    env.close()


def test_refresh_matches_set_goal_and_obstacle_semantics(oracle):
    env = make(oracle, _abi.ENV_DYN, n=8, seed=4)
    env.buf["goal"][:, 0] = [0.5, 0.1, 0.1, -2.0, 0.0, -1.0]
    env.buf["obst_start"][:, 0] = [0.8, -0.5, 0.5, 1.0, 1.0, 0.0]
    env.buf["obst_end"][:, 0] = [0.9, 0.6, 0.4, -1.0, -2.0, 0.0]
    mask = np.zeros(8, np.uint8)
    mask[0] = 1
    other = env.buf["observation"][1].copy()
    env.refresh(mask)
    assert np.all(env.buf["observation"][1] == other)
    assert np.max(np.abs(env.buf["obst_pos"][:, 0] - [0.8, -0.5, 0.5])) == 0
    assert np.max(np.abs(env.buf["obst_vel"][:6, 0] - oracle.dyn_velocity(env.buf["obst_start"][:, 0], env.buf["obst_end"][:, 0]))) < 1e-15
    assert np.all(env.buf["observation"][0, 12:18] == env.buf["goal"][:, 0].astype(np.float32))
    env.close()


def test_demo_style_single_env_loop(oracle):
    """BASELINE.json configs[0]: UR5OriReach-v1, 1 env, demo.py-style random loop (demo.py:8-15) — plumbing."""
    env = make(oracle, _abi.ENV_ORI, n=1, seed=0)
    rng = np.random.default_rng(0)
    resets = 0
    for _ in range(300):
        env.step(rng.uniform(-1, 1, (1, 6)).astype(np.float32))
        assert np.isfinite(env.buf["reward"]).all() and np.all(np.abs(env.buf["observation"]) <= 10.0)
        resets += int(env.buf["terminated"][0] or env.buf["truncated"][0])
    assert resets >= 3 and env.buf["episode_id"][0] == resets + 1
    env.close()


def test_guided_gjk_start_stays_within_path_tolerance(oracle):
    """urgym_config.gjk_start = GUIDED is an opt-in search start, not the reference's: over random arm / obstacle poses
    it is NOT inside the 1e-4 m parity tolerance on every query (Bullet's exits return the current iterate, so the answer
    depends on the search path at the 1e-5 .. 1e-4 level on these finely faceted hulls) and is therefore never the
    default.  Pinned here: the size of that deviation — < 5e-4 m always, > 1e-5 m on under 0.5 % of queries, > 1e-6 m on
    under 3 % — and that the collision verdict only moves at knife-edge distances."""
    rng = np.random.default_rng(11)
    lo = np.array([-2 * np.pi, -2 * np.pi, -np.pi, -2 * np.pi, -2 * np.pi, -2 * np.pi])
    worst, over6, over5, flips, n = 0.0, 0, 0, 0, 3000
    for _ in range(n):
        q = rng.uniform(lo, -lo) * 0.5
        quat = Rot.random(random_state=int(rng.integers(1 << 30))).as_quat()
        pose = np.r_[rng.uniform([-0.1, -0.6, 0.05], [0.9, 0.6, 0.9]), quat]
        ld0, c0, _ = oracle.query(q, pose, gjk_start=_abi.GJK_START_BULLET)
        ld1, c1, _ = oracle.query(q, pose, gjk_start=_abi.GJK_START_GUIDED)
        d = np.abs(ld0 - ld1)
        worst = max(worst, float(d.max()))
        over6 += int((d > 1e-6).sum())
        over5 += int((d > 1e-5).sum())
        flips += int(c0 != c1)
    assert worst < 5e-4, worst
    assert over6 < 0.03 * 5 * n and over5 < 0.005 * 5 * n, (over6, over5)
    assert flips <= 3, flips


# ------------------------------------------------------------------------------------------------ penetration depth (EPA)
def test_penetration_depth_analytic_cases(oracle):
    """Overlapping cores: the oracle reports -(depth of the margin-inflated shapes) like p.getClosestPoints' contact distance
    (pyb_setup.py:452 stores it): depth(inflated) = depth(cores) + margin_A + margin_B."""
    cyl = [0.05, 0.4]
    at = lambda x, y, z: [x, y, z, *IDENT]
    # sphere r 0.02 inside the obstacle cylinder (r 0.05, h 0.4), 3 cm off the axis: nearest exit is radial, 2 cm + the sphere
    r = oracle.closest(oracle.SPHERE, [0.02], at(0.03, 0, 0), oracle.CYLZ, cyl, at(0, 0, 0))
    assert r["penetrating"] and abs(r["distance"] - (-(0.02 + 0.02))) < 1e-8
    # ... and 1 cm below the top cap on the axis: nearest exit is axial
    r = oracle.closest(oracle.SPHERE, [0.02], at(0, 0, 0.19), oracle.CYLZ, cyl, at(0, 0, 0))
    assert r["penetrating"] and abs(r["distance"] - (-(0.01 + 0.02))) < 1e-8
    # two boxes of half 0.1 overlapping by 5 cm along x (face contact): depth 0.05
    r = oracle.closest(oracle.BOX, [0.1] * 3, at(0, 0, 0), oracle.BOX, [0.1] * 3, at(0.15, 0.02, 0.01))
    assert r["penetrating"] and abs(r["distance"] - (-0.05)) < 1e-8
    # a 5 cm cube sunk 3 cm (centre) below the table top (z = -0.12): it has to rise by 0.05 + 0.03
    r = oracle.closest(oracle.BOX, [0.05] * 3, at(0.5, 0, -0.15), oracle.BOX, [0.55, 0.9, 0.46], at(0.5, 0, -0.58))
    assert r["penetrating"] and abs(r["distance"] - (-0.08)) < 1e-8
    # rigid-motion invariance of a hull <-> cylinder depth
    rng = np.random.default_rng(3)
    pa = np.r_[0.4, 0.1, 0.3, Rot.random(random_state=1).as_quat()]
    pb = np.r_[0.41, 0.12, 0.33, Rot.random(random_state=2).as_quat()]
    d0 = oracle.closest(oracle.HULL, [3], pa, oracle.CYLZ, cyl, pb)
    assert d0["penetrating"] and d0["distance"] < -0.002
    G = Rot.random(random_state=5)
    t = rng.uniform(-1, 1, 3)
    mv = lambda p: np.r_[G.apply(p[:3]) + t, (G * Rot.from_quat(p[3:])).as_quat()]
    d1 = oracle.closest(oracle.HULL, [3], mv(pa), oracle.CYLZ, cyl, mv(pb))
    assert d1["penetrating"] and abs(d1["distance"] - d0["distance"]) < 1e-7


def test_penetration_depth_is_a_lower_envelope_of_the_support_function(oracle):
    """depth(cores) = min over unit n of h_{A-B}(n): no sampled direction may give a smaller value than the EPA result, and a
    local search started at the best sample must not get below it either (hull <-> cylinder, the pair of get_link_distances)."""
    verts = [MODEL["hull_verts"][MODEL["hull_offset"][l - 1]:MODEL["hull_offset"][l]] for l in range(1, 7)]
    rng = np.random.default_rng(8)
    checked = 0
    for _ in range(40):
        link = int(rng.integers(2, 7))
        Ra = Rot.random(random_state=int(rng.integers(1 << 30)))
        ta = rng.uniform(-0.2, 0.2, 3)
        Rb = Rot.random(random_state=int(rng.integers(1 << 30)))
        tb = ta + Ra.apply(verts[link - 1].mean(0)) + rng.normal(0, 0.02, 3)
        got = oracle.closest(oracle.HULL, [link], np.r_[ta, Ra.as_quat()], oracle.CYLZ, [0.05, 0.4], np.r_[tb, Rb.as_quat()])
        if not got["penetrating"]:
            continue
        depth = -got["distance"] - 0.002  # cores
        Vw = Ra.apply(verts[link - 1]) + ta

        def h(n):  # support of core_A - core_B in direction n (world frame)
            n = n / np.linalg.norm(n)
            nb = Rb.inv().apply(-n)
            s = np.hypot(nb[0], nb[1])
            pb = np.array([0.049 * nb[0] / s, 0.049 * nb[1] / s, 0.199 * np.sign(nb[2])]) if s > 0 else np.array([0.049, 0, 0.199 * np.sign(nb[2])])
            return (Vw @ n).max() - (Rb.apply(pb) + tb) @ n

        dirs = rng.normal(size=(4000, 3))
        vals = np.array([h(d) for d in dirs])
        assert vals.min() >= depth - 1e-9
        # h is piecewise smooth and not convex on the sphere: polish several of the best samples and keep the lowest
        best = min(minimize(h, dirs[i], method="Nelder-Mead", options={"xatol": 1e-9, "fatol": 1e-12, "maxiter": 3000}).fun
                   for i in np.argsort(vals)[:10])
        assert best >= depth - 1e-8 and best - depth < 3e-4, (best, depth)   # nothing below the EPA value, and it is attained
        checked += 1
    assert checked >= 15
