#!/usr/bin/env python3
"""Generate the UR5e kinematic/collision data tables used by the HIP path and the oracle.

Run ONCE in the build container (needs /root/reference, which never travels to the GPU box):

    python tools/gen_model.py [--reference /root/reference]

Inputs (data files of the reference, read as data):
  UR_gym/envs/robots/urdf/ur5e.urdf                      joint origins (urdf:232-279), collision origins (urdf:75-213)
  UR_gym/envs/robots/meshes/ur5/collision/{shoulder,upperarm,forearm,wrist1,wrist2,wrist3}.stl

Outputs (committed, derived numeric tables only):
  data/ur5e_model.h     C arrays (double) included by ur_gym_amd/csrc and oracle/
  data/ur5e_model.npz   the same tables for Python-side tests

What is modelled:
  * joint k fixed transform  T_k = Trans(xyz_k) * Rz(yaw) * Ry(pitch) * Rx(roll)   (URDF fixed-axis rpy)
  * collision hull k (PyBullet link index k = 1..6) = the STL vertex cloud, passed through the integer-grid
    quantisation that Bullet's btConvexHullShape::optimizeConvexHull() (btConvexHullComputer, 10216 grid steps
    per axis, truncation toward the cloud centre) applies when a URDF convex mesh is imported
    [UNVERIFIED-BULLET: restated from the published bullet3 algorithm; pybullet is not installable here], then
    moved into the link frame by the <collision><origin>.
  * a bounding capsule per hull (segment + radius) used ONLY for conservative culling on the GPU.
"""
import argparse
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

LINKS = [  # (pybullet link index, urdf link name, stl)
    (1, "shoulder_link", "shoulder.stl"),
    (2, "upper_arm_link", "upperarm.stl"),
    (3, "forearm_link", "forearm.stl"),
    (4, "wrist_1_link", "wrist1.stl"),
    (5, "wrist_2_link", "wrist2.stl"),
    (6, "wrist_3_link", "wrist3.stl"),
]
JOINTS = ["shoulder_pan_joint", "shoulder_lift_joint", "elbow_joint", "wrist_1_joint", "wrist_2_joint", "wrist_3_joint"]


def rpy_matrix(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def read_stl_vertices(path):
    raw = open(path, "rb").read()
    ntri = struct.unpack("<I", raw[80:84])[0]
    assert len(raw) == 84 + 50 * ntri, "not a binary STL"
    rec = np.frombuffer(raw[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    return rec["v"].reshape(-1, 3).astype(np.float64)  # float32 file values promoted exactly


def bullet_hull_quantise(pts):
    """btConvexHullInternal::compute + getCoordinates, restated (double-precision build).

    Returns the de-quantised vertex cloud (duplicates removed). Interior points are harmless for a support
    function; they are stripped afterwards with Qhull.
    """
    mn, mx = pts.min(0), pts.max(0)
    s = mx - mn
    max_axis = (2 if s[1] < s[2] else 1) if s[0] < s[1] else (2 if s[0] < s[2] else 0)
    min_axis = (0 if s[0] < s[2] else 2) if s[0] < s[1] else (1 if s[1] < s[2] else 2)
    if min_axis == max_axis:
        min_axis = (max_axis + 1) % 3
    med_axis = 3 - max_axis - min_axis
    s = s / 10216.0
    if ((med_axis + 1) % 3) != max_axis:
        s = -s
    scaling = s.copy()
    inv = np.where(s != 0, 1.0 / np.where(s != 0, s, 1.0), 0.0)
    center = (mn + mx) * 0.5
    q = np.trunc((pts - center) * inv)  # (int32_t) cast truncates toward zero
    q = np.unique(q, axis=0)
    return q * scaling + center


def _seg_radius(v, p0, p1):
    d = p1 - p0
    dd = d @ d
    u = np.clip(((v - p0) @ d) / dd, 0, 1) if dd > 1e-18 else np.zeros(len(v))
    closest = p0 + u[:, None] * d
    return np.sqrt(((v - closest) ** 2).sum(1)).max()


def bounding_capsule(v):
    """Tight bounding capsule: minimise the capsule volume over the two segment end points."""
    from scipy.optimize import minimize
    c = 0.5 * (v.min(0) + v.max(0))
    _, _, vt = np.linalg.svd(v - v.mean(0), full_matrices=False)
    t = (v - c) @ vt[0]
    best = None
    for shrink in (0.0, 0.03, 0.06, 0.09):
        x0 = np.concatenate([c + (t.min() + shrink) * vt[0], c + (t.max() - shrink) * vt[0]])
        def vol(x):
            r = _seg_radius(v, x[:3], x[3:])
            return np.pi * r * r * np.linalg.norm(x[3:] - x[:3]) + 4.0 / 3.0 * np.pi * r ** 3

        res = minimize(vol, x0, method="Nelder-Mead",
                       options={"xatol": 1e-6, "fatol": 1e-10, "maxiter": 20000, "maxfev": 20000})
        if best is None or res.fun < best.fun:
            best = res
    p0, p1 = best.x[:3], best.x[3:]
    rad = _seg_radius(v, p0, p1) * (1 + 1e-9) + 1e-9  # conservative by construction
    return p0, p1, rad


def hull_adjacency(v):
    """Vertex adjacency of the convex hull surface (Qhull triangulation, option Qt): every polytope edge is an edge of
    the triangulation, so a vertex none of whose neighbours has a larger d.x is a global maximiser of d.x."""
    hull = ConvexHull(v)
    assert len(hull.vertices) == len(v), "every table vertex must be a hull vertex"
    nbr = [set() for _ in range(len(v))]
    for tri in hull.simplices:
        for a in range(3):
            for b in range(3):
                if a != b:
                    nbr[tri[a]].add(int(tri[b]))
    return [sorted(s) for s in nbr]


def farthest_point_seeds(v, k):
    idx = [int(np.argmax(np.linalg.norm(v - v.mean(0), axis=1)))]
    d = np.linalg.norm(v - v[idx[0]], axis=1)
    while len(idx) < k:
        i = int(np.argmax(d))
        idx.append(i)
        d = np.minimum(d, np.linalg.norm(v - v[i], axis=1))
    return idx


def climb(v, nbr, seeds, d):
    cur = seeds[int(np.argmax(v[seeds] @ d))]
    best = v[cur] @ d
    steps = 0
    while True:
        cand = nbr[cur]
        t = v[cand] @ d
        j = int(np.argmax(t))
        if t[j] > best:
            best, cur = t[j], cand[j]
            steps += 1
        else:
            return cur, steps


def fmt(x):
    return repr(float(x))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "data"))
    args = ap.parse_args()
    urdf = os.path.join(args.reference, "UR_gym/envs/robots/urdf/ur5e.urdf")
    mesh_dir = os.path.join(args.reference, "UR_gym/envs/robots/meshes/ur5/collision")
    root = ET.parse(urdf).getroot()
    joints = {j.get("name"): j for j in root.findall("joint")}
    links = {l.get("name"): l for l in root.findall("link")}

    jxyz, jrpy, jrot, jlim = [], [], [], []
    for name in JOINTS:
        j = joints[name]
        o = j.find("origin")
        xyz = np.array([float(t) for t in o.get("xyz").split()])
        rpy = np.array([float(t) for t in o.get("rpy").split()])
        assert [float(t) for t in j.find("axis").get("xyz").split()] == [0.0, 0.0, 1.0]
        lim = j.find("limit")
        jxyz.append(xyz)
        jrpy.append(rpy)
        jrot.append(rpy_matrix(*rpy))
        jlim.append([float(lim.get("lower")), float(lim.get("upper"))])
    # ee_fixed_joint must be the identity for "link 7 frame == wrist_3 frame" (urdf:294-298)
    eo = joints["ee_fixed_joint"].find("origin")
    assert all(float(t) == 0.0 for t in (eo.get("xyz") + " " + eo.get("rpy")).split())

    hull_off, hull_verts, caps, raw_counts = [0], [], [], []
    adj_lists, seeds_all = [], []
    NSEED = 16
    rng = np.random.default_rng(0)
    for idx, lname, stl in LINKS:
        col = links[lname].find("collision")
        o = col.find("origin")
        oxyz = np.array([float(t) for t in o.get("xyz").split()])
        orpy = np.array([float(t) for t in o.get("rpy").split()])
        pts = np.unique(read_stl_vertices(os.path.join(mesh_dir, stl)), axis=0)
        raw_counts.append(len(pts))
        qpts = bullet_hull_quantise(pts)
        hv = qpts[np.sort(ConvexHull(qpts).vertices)]
        # sanity: quantisation moves a vertex by less than one grid step per axis
        step = (pts.max(0) - pts.min(0)) / 10216.0
        assert np.all(np.abs(hv.max(0) - pts.max(0)) <= step + 1e-12)
        v_link = hv @ rpy_matrix(*orpy).T + oxyz
        hull_verts.append(v_link)
        hull_off.append(hull_off[-1] + len(v_link))
        caps.append(bounding_capsule(v_link))
        nbr = hull_adjacency(v_link)
        seeds = farthest_point_seeds(v_link, NSEED)
        # offline proof-by-test of the hill-climbing support: equals the brute-force arg-max for random directions
        worst_steps, tot = 0, 0
        for _ in range(20000):
            d = rng.normal(size=3)
            cur, steps = climb(v_link, nbr, seeds, d)
            assert v_link[cur] @ d >= (v_link @ d).max() - 1e-15 * np.linalg.norm(d), "hill climbing missed the support vertex"
            worst_steps = max(worst_steps, steps)
            tot += steps
        deg = [len(x) for x in nbr]
        print(f"   adjacency: degree mean {np.mean(deg):.1f} max {max(deg)}; climb steps from {NSEED} seeds: mean {tot/20000:.2f} max {worst_steps}")
        base = hull_off[-2]
        adj_lists += [[base + j for j in x] for x in nbr]
        seeds_all.append([base + j for j in seeds])
        print(f"link {idx} {lname}: stl unique {len(pts)} -> bullet-quantised hull {len(v_link)} verts, "
              f"capsule r={caps[-1][2]:.4f} len={np.linalg.norm(caps[-1][1]-caps[-1][0]):.4f}")
    allv = np.concatenate(hull_verts)
    adj_off = np.cumsum([0] + [len(x) for x in adj_lists]).astype(np.int32)
    adj_idx = np.array([j for x in adj_lists for j in x], dtype=np.uint16)

    os.makedirs(args.out, exist_ok=True)
    np.savez(os.path.join(args.out, "ur5e_model.npz"),
             joint_xyz=np.array(jxyz), joint_rpy=np.array(jrpy), joint_rot=np.array(jrot), joint_limits=np.array(jlim),
             hull_offset=np.array(hull_off, dtype=np.int32), hull_verts=allv,
             capsule_p0=np.array([c[0] for c in caps]), capsule_p1=np.array([c[1] for c in caps]),
             capsule_r=np.array([c[2] for c in caps]), adj_offset=adj_off, adj_index=adj_idx,
             seeds=np.array(seeds_all, dtype=np.int32))

    with open(os.path.join(args.out, "ur5e_model.h"), "w") as f:
        f.write("/* GENERATED by tools/gen_model.py from the reference's ur5e.urdf and collision STLs (data tables only).\n"
                " * Joint k: T_k = Trans(UR5E_JOINT_XYZ[k]) * UR5E_JOINT_ROT[k] * Rz(q_k)   (ur5e.urdf:232-279)\n"
                " * Hull of PyBullet link L (1..6): UR5E_HULL_VERTS[UR5E_HULL_OFFSET[L-1] .. UR5E_HULL_OFFSET[L]) in the LINK frame\n"
                " *   (collision <origin> baked in, ur5e.urdf:75-213; Bullet optimizeConvexHull grid quantisation applied).\n"
                " * Do not edit. */\n#ifndef UR5E_MODEL_H\n#define UR5E_MODEL_H\n\n")
        f.write(f"#define UR5E_NUM_JOINTS 6\n#define UR5E_NUM_HULLS 6\n#define UR5E_NUM_HULL_VERTS {len(allv)}\n\n")
        f.write("static const double UR5E_JOINT_XYZ[6][3] = {\n" + ",\n".join(
            "  {" + ", ".join(fmt(x) for x in r) + "}" for r in jxyz) + "\n};\n")
        f.write("static const double UR5E_JOINT_ROT[6][9] = {\n" + ",\n".join(
            "  {" + ", ".join(fmt(x) for x in r.reshape(-1)) + "}" for r in jrot) + "\n};\n")
        f.write("static const double UR5E_JOINT_LIMITS[6][2] = {\n" + ",\n".join(
            "  {" + ", ".join(fmt(x) for x in r) + "}" for r in jlim) + "\n};\n")
        f.write("static const int UR5E_HULL_OFFSET[7] = {" + ", ".join(str(x) for x in hull_off) + "};\n")
        f.write("static const double UR5E_CAPSULE[6][7] = { /* p0.xyz, p1.xyz, radius (link frame) */\n" + ",\n".join(
            "  {" + ", ".join(fmt(x) for x in list(c[0]) + list(c[1]) + [c[2]]) + "}" for c in caps) + "\n};\n")
        f.write(f"#define UR5E_NUM_ADJ {len(adj_idx)}\n#define UR5E_NUM_SEEDS {NSEED}\n")
        f.write("/* hull surface graph (CSR over GLOBAL vertex ids): neighbours of vertex i = UR5E_ADJ_INDEX[UR5E_ADJ_OFFSET[i] .. UR5E_ADJ_OFFSET[i+1]) */\n")
        f.write(f"static const int UR5E_ADJ_OFFSET[{len(adj_off)}] = {{" + ", ".join(str(int(x)) for x in adj_off) + "};\n")
        f.write(f"static const unsigned short UR5E_ADJ_INDEX[{len(adj_idx)}] = {{" + ", ".join(str(int(x)) for x in adj_idx) + "};\n")
        f.write("/* well-spread start vertices (global ids) for the hill-climbing support search */\n")
        f.write(f"static const unsigned short UR5E_SEEDS[6][{NSEED}] = {{\n" + ",\n".join(
            "  {" + ", ".join(str(int(x)) for x in r) + "}" for r in seeds_all) + "\n};\n")
        f.write(f"static const double UR5E_HULL_VERTS[{len(allv)}][3] = {{\n")
        f.write(",\n".join("  {" + ", ".join(fmt(x) for x in r) + "}" for r in allv))
        f.write("\n};\n\n#endif\n")
    print("total hull vertices:", len(allv), "(raw STL unique:", sum(raw_counts), ")")


if __name__ == "__main__":
    main()
