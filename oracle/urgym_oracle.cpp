/*
 * urgym_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, double-precision restatement of the reference hot path (WanqingXia/UR-gym: RobotTaskEnv.step /
 * reset for UR5OriReach-v1, UR5ObsReach-v1, UR5DynReach-v1) used as the parity checker for the HIP path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (ur_gym_amd/) never does.
 *
 * PARITY STATUS: pinned at the pybullet boundary by outputs of the reference's own PyBullet environments, as far as the
 * reference holds any.  pybullet is not installed/installable here and the reference ships no tests, but its SAC checkpoints
 * carry two consecutive observations each as plain text in their JSON (tests/golden/reference_observations.json, generator
 * next to it).  Against them (tests/test_reference_pins.py): forward kinematics + Bullet's Euler read-out (8 samples,
 * <= 7e-7 m), getClosestPoints link distances incl. collision margins (25 values, <= 1e-7 m), the moving obstacle's
 * integration (3e-8 m) and one whole step() per env.  utils.distance / utils.angular_distance are pinned by fixtures generated
 * from the reference's own UR_gym/utils.py (tests/golden/utils_golden.json); the closed-loop replay of the four shipped actors
 * reproduces the per-trial statistics the reference ships (tests/test_closed_loop.py).
 * STILL UNPINNED (no reference value exists): the path-dependent last digits of Bullet's GJK at degenerate simplices, hull <->
 * hull self-collision verdicts, the exact value of a penetration depth (Bullet's own EPA accuracy is 1e-4), the Euler branches at
 * |sin pitch| >= 0.99999, physics side effects of stepSimulation on the teleported arm.  Everything tagged [BULLET] below restates
 * the published bullet3 algorithm (btGjkPairDetector, btVoronoiSimplexSolver, btMultiBody, pybullet.c quaternion helpers) from its
 * call sites in UR_gym/pyb_setup.py (pybullet is a dependency of the reference, setup.py:22, version unpinned).
 *
 * Every function cites the reference file:line (under /root/reference) it follows.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <utility>
#include <algorithm>

#include "../include/urgym.h"
#include "../data/ur5e_model.h"

namespace {

// ------------------------------------------------------------------------------------------------ small math
struct V3 {
  double x, y, z;
};
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline double len2(V3 a) { return dot(a, a); }

struct M3 {
  double m[3][3];
};
inline M3 m3_identity() { return M3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
inline M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
inline V3 mul(const M3& a, V3 v) {
  return v3(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}
inline V3 mulT(const M3& a, V3 v) {  // a^T v
  return v3(a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z, a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
            a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z);
}
struct X3 {  // rigid transform
  M3 R;
  V3 t;
};
struct Quat {
  double x, y, z, w;
};
inline Quat qmul(Quat a, Quat b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Quat qnormalize(Quat q) {
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / n, q.y / n, q.z / n, q.w / n};
}
inline M3 quat_to_mat(Quat q) {  // btMatrix3x3::setRotation
  double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  double s = 2.0 / d;
  double xs = q.x * s, ys = q.y * s, zs = q.z * s;
  double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  return M3{{{1.0 - (yy + zz), xy - wz, xz + wy}, {xy + wz, 1.0 - (xx + zz), yz - wx}, {xz - wy, yz + wx, 1.0 - (xx + yy)}}};
}

// [BULLET] btMatrix3x3::getRotation — how getLinkState turns the cached link basis into the quaternion that
// PyBullet.get_link_orientation (pyb_setup.py:234-253) feeds to getEulerFromQuaternion.
inline Quat mat_to_quat(const M3& a) {
  double trace = a.m[0][0] + a.m[1][1] + a.m[2][2];
  double t[4];
  if (trace > 0.0) {
    double s = std::sqrt(trace + 1.0);
    t[3] = s * 0.5;
    s = 0.5 / s;
    t[0] = (a.m[2][1] - a.m[1][2]) * s;
    t[1] = (a.m[0][2] - a.m[2][0]) * s;
    t[2] = (a.m[1][0] - a.m[0][1]) * s;
  } else {
    int i = a.m[0][0] < a.m[1][1] ? (a.m[1][1] < a.m[2][2] ? 2 : 1) : (a.m[0][0] < a.m[2][2] ? 2 : 0);
    int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = std::sqrt(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0);
    t[i] = s * 0.5;
    s = 0.5 / s;
    t[3] = (a.m[k][j] - a.m[j][k]) * s;
    t[j] = (a.m[j][i] + a.m[i][j]) * s;
    t[k] = (a.m[k][i] + a.m[i][k]) * s;
  }
  return Quat{t[0], t[1], t[2], t[3]};
}

// [BULLET] pybullet getQuaternionFromEuler (pyb_setup.py:151-152, 313-314): q = qz(yaw) * qy(pitch) * qx(roll).
inline Quat quat_from_euler_bullet(double roll, double pitch, double yaw) {
  double phi = roll * 0.5, the = pitch * 0.5, psi = yaw * 0.5;
  double sp = std::sin(phi), cp = std::cos(phi), st = std::sin(the), ct = std::cos(the), ss = std::sin(psi), cs = std::cos(psi);
  return Quat{sp * ct * cs - cp * st * ss, cp * st * cs + sp * ct * ss, cp * ct * ss - sp * st * cs, cp * ct * cs + sp * st * ss};
}

// [BULLET] pybullet getEulerFromQuaternion (pyb_setup.py:190, 248): (roll, pitch, yaw), R = Rz(yaw)Ry(pitch)Rx(roll),
// with the |sin(pitch)| >= 0.99999 gimbal branches.
inline void euler_from_quat_bullet(Quat q, double rpy[3]) {
  double sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
  double sarg = -2.0 * (q.x * q.z - q.w * q.y);
  if (sarg <= -0.99999) {
    rpy[0] = 0;
    rpy[1] = -0.5 * M_PI;
    rpy[2] = 2 * std::atan2(q.x, -q.y);
  } else if (sarg >= 0.99999) {
    rpy[0] = 0;
    rpy[1] = 0.5 * M_PI;
    rpy[2] = 2 * std::atan2(-q.x, q.y);
  } else {
    rpy[0] = std::atan2(2 * (q.y * q.z + q.w * q.x), squ - sqx - sqy + sqz);
    rpy[1] = std::asin(sarg);
    rpy[2] = std::atan2(2 * (q.x * q.y + q.w * q.z), squ + sqx - sqy - sqz);
  }
}

// [BULLET] pybullet getDifferenceQuaternion (pyb_setup.py:351-359): dq = nearest(q1 to q0) * q0^-1.
inline Quat quat_difference_bullet(Quat q0, Quat q1) {
  double dm = (q0.x - q1.x) * (q0.x - q1.x) + (q0.y - q1.y) * (q0.y - q1.y) + (q0.z - q1.z) * (q0.z - q1.z) + (q0.w - q1.w) * (q0.w - q1.w);
  double dp = (q0.x + q1.x) * (q0.x + q1.x) + (q0.y + q1.y) * (q0.y + q1.y) + (q0.z + q1.z) * (q0.z + q1.z) + (q0.w + q1.w) * (q0.w + q1.w);
  Quat q1n = (dm < dp) ? q1 : Quat{-q1.x, -q1.y, -q1.z, -q1.w};
  Quat q0inv{-q0.x, -q0.y, -q0.z, q0.w};
  return qmul(q1n, q0inv);
}
// [BULLET] pybullet getAxisAngleFromQuaternion (pyb_setup.py:361-363): angle = 2 acos(w), axis = xyz / sqrt(1-w^2).
inline void axis_angle_bullet(Quat q, V3* axis, double* angle) {
  double w = std::min(1.0, std::max(-1.0, q.w));
  *angle = 2.0 * std::acos(w);
  double s2 = 1.0 - q.w * q.w;
  if (s2 < 10.0 * 2.220446049250313e-16) {
    *axis = v3(1, 0, 0);
  } else {
    double s = 1.0 / std::sqrt(s2);
    *axis = v3(q.x * s, q.y * s, q.z * s);
  }
}

// scipy Rotation.from_euler('ZYX', [a0,a1,a2]).as_quat() as used by utils.angular_distance (utils.py:47-55):
// intrinsic Z-Y'-X'' => q = qz(a0) * qy(a1) * qx(a2).  The reference feeds (roll,pitch,yaw) as (a0,a1,a2); copied literally.
inline Quat quat_scipy_ZYX(double a0, double a1, double a2) {
  Quat qz{0, 0, std::sin(a0 * 0.5), std::cos(a0 * 0.5)};
  Quat qy{0, std::sin(a1 * 0.5), 0, std::cos(a1 * 0.5)};
  Quat qx{std::sin(a2 * 0.5), 0, 0, std::cos(a2 * 0.5)};
  return qmul(qmul(qz, qy), qx);
}
// utils.distance (utils.py:5-31)
inline double pos_distance(const double* a, const double* b) {
  double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return std::sqrt(dx * dx + dy * dy + dz * dz);
}
// utils.angular_distance (utils.py:34-69)
inline double angular_distance(const double* a, const double* b) {
  Quat qa = quat_scipy_ZYX(a[3], a[4], a[5]), qb = quat_scipy_ZYX(b[3], b[4], b[5]);
  double d = qa.x * qb.x + qa.y * qb.y + qa.z * qb.z + qa.w * qb.w;
  d = std::min(1.0, std::max(-1.0, d));
  return 2.0 * std::acos(std::fabs(d));
}

// ------------------------------------------------------------------------------------------------ kinematics
// Forward kinematics of the ur5e.urdf chain (urdf:232-279): link[k], k=1..6, is the world
// frame of PyBullet link index k; link 7 (ee_link, UR5.py:263) coincides with link 6 (urdf:294-298).
// Base at the world origin with identity orientation (ur_tasks.py:88, core.py:51-56 useFixedBase).
void forward_kinematics(const double q[6], X3 link[7]) {
  X3 T{m3_identity(), v3(0, 0, 0)};
  link[0] = T;
  for (int k = 0; k < 6; k++) {
    M3 F;
    std::memcpy(F.m, UR5E_JOINT_ROT[k], sizeof(F.m));
    V3 o = v3(UR5E_JOINT_XYZ[k][0], UR5E_JOINT_XYZ[k][1], UR5E_JOINT_XYZ[k][2]);
    double c = std::cos(q[k]), s = std::sin(q[k]);
    M3 Rz{{{c, -s, 0}, {s, c, 0}, {0, 0, 1}}};
    T.t = T.t + mul(T.R, o);
    T.R = mul(mul(T.R, F), Rz);
    link[k + 1] = T;
  }
}

// ------------------------------------------------------------------------------------------------ shapes
// [BULLET] collision shapes as pybullet builds them:
//  HULL : btConvexHullShape of the STL vertices, core = the hull itself, margin = 0.001 (URDF default margin)
//  CYLZ : btCylinderShapeZ(radius,radius,height/2) made by p.createCollisionShape (pyb_setup.py:748-752):
//         PhysicsServerCommandProcessor ends createCollisionShape with shape->setMargin(m_defaultCollisionMargin = 0.001),
//         and btCylinderShape::setMargin re-derives the implicit dimensions: core = (r - 0.001, h/2 - 0.001), margin 0.001.
//         PINNED by the reference's own data: the link_dist slots of the observations stored in
//         Trained_Models/*/best_model.zip (tests/golden/reference_observations.json) are reproduced to 1e-7 with
//         0.001 and are off by up to 1.7e-3 with the constructor's "safe margin" (0.005), tests/test_reference_pins.py.
//  BOX  : btBoxShape(half) through the same createCollisionShape: core = half - 0.001, margin 0.001 (pinned likewise
//         by the table / track distances of the Sep-2023 checkpoints' observations).
//  POINT: btSphereShape(r): core = centre point, margin = r (btSphereShape::getMargin returns the radius)
enum ShapeType { SH_HULL = 0, SH_CYLZ = 1, SH_BOX = 2, SH_POINT = 3 };
struct Shape {
  int type;
  const double (*verts)[3];
  int nverts;
  double hx, hy, hz;  // core half dims (cyl: hx = core radius, hz = core half height)
  double margin;
  X3 pose;
};
const double PRIMITIVE_MARGIN = 0.001;  // [BULLET] PhysicsServerCommandProcessor::m_defaultCollisionMargin
// What-if switch (tests only): >= 0 replaces the margin of cylinders / boxes, e.g. 0.005 = the "safe margin" of the
// btCylinderShape constructor that round 1 assumed; tests/test_reference_pins.py shows that it misses the reference.
double g_prim_margin_override = -1.0;
inline double primitive_margin() { return g_prim_margin_override >= 0 ? g_prim_margin_override : PRIMITIVE_MARGIN; }
Shape make_hull(int link /*1..6*/, const X3& pose) {
  Shape s{};
  s.type = SH_HULL;
  s.verts = &UR5E_HULL_VERTS[UR5E_HULL_OFFSET[link - 1]];
  s.nverts = UR5E_HULL_OFFSET[link] - UR5E_HULL_OFFSET[link - 1];
  s.margin = 0.001;
  s.pose = pose;
  return s;
}
Shape make_box(double hx, double hy, double hz, const X3& pose) {
  Shape s{};
  s.type = SH_BOX;
  s.margin = primitive_margin();
  s.hx = hx - s.margin;
  s.hy = hy - s.margin;
  s.hz = hz - s.margin;
  s.pose = pose;
  return s;
}
Shape make_cylinder_z(double radius, double height, const X3& pose) {
  Shape s{};
  s.type = SH_CYLZ;
  s.margin = primitive_margin();
  s.hx = s.hy = radius - s.margin;
  s.hz = 0.5 * height - s.margin;
  s.pose = pose;
  return s;
}
Shape make_sphere(double radius, const X3& pose) {
  Shape s{};
  s.type = SH_POINT;
  s.margin = radius;
  s.pose = pose;
  return s;
}
// What-if switch for precision studies of the HIP path (tests only): bit0 = hull vertices rounded to float32,
// bit1 = support scan evaluated in float32 (direction cast, fmaf chain z,y,x as on the device).
thread_local int g_emulate = 0;

// [BULLET] btConvexShape::localGetSupportVertexWithoutMarginNonVirtual
V3 support_local(const Shape& s, V3 d) {
  switch (s.type) {
    case SH_HULL: {
      if (g_emulate) {
        int bi = 0;
        if (g_emulate & 2) {
          float best = -3.0e38f, dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
          for (int i = 0; i < s.nverts; i++) {
            float v = std::fmaf((float)s.verts[i][2], dz, std::fmaf((float)s.verts[i][1], dy, (float)s.verts[i][0] * dx));
            if (v > best) { best = v; bi = i; }
          }
        } else {
          double best = -1e300;
          for (int i = 0; i < s.nverts; i++) {
            double v = (double)(float)s.verts[i][0] * d.x + (double)(float)s.verts[i][1] * d.y + (double)(float)s.verts[i][2] * d.z;
            if (v > best) { best = v; bi = i; }
          }
        }
        if (g_emulate & 1) return v3((double)(float)s.verts[bi][0], (double)(float)s.verts[bi][1], (double)(float)s.verts[bi][2]);
        return v3(s.verts[bi][0], s.verts[bi][1], s.verts[bi][2]);
      }
      double best = -1e300;
      int bi = 0;
      for (int i = 0; i < s.nverts; i++) {
        double v = s.verts[i][0] * d.x + s.verts[i][1] * d.y + s.verts[i][2] * d.z;
        if (v > best) {
          best = v;
          bi = i;
        }
      }
      return v3(s.verts[bi][0], s.verts[bi][1], s.verts[bi][2]);
    }
    case SH_CYLZ: {
      double r = s.hx, h = s.hz;
      double sn = std::sqrt(d.x * d.x + d.y * d.y);
      if (sn != 0.0) {
        double k = r / sn;
        return v3(d.x * k, d.y * k, d.z < 0.0 ? -h : h);
      }
      return v3(r, 0.0, d.z < 0.0 ? -h : h);
    }
    case SH_BOX:
      return v3(d.x >= 0 ? s.hx : -s.hx, d.y >= 0 ? s.hy : -s.hy, d.z >= 0 ? s.hz : -s.hz);
    default:
      return v3(0, 0, 0);
  }
}
inline V3 support_world(const Shape& s, V3 dir_world) {
  V3 l = support_local(s, mulT(s.pose.R, dir_world));
  return mul(s.pose.R, l) + s.pose.t;
}

// ------------------------------------------------------------------------------------------------ GJK
// [BULLET] btVoronoiSimplexSolver restated (closest point of a <=4-vertex simplex to the origin, with vertex
// reduction).  Only the Minkowski-difference vertices w are kept: the path needs distances, not witness points.
struct Simplex {
  int n = 0;
  V3 w[4];
  V3 last_w{1e300, 1e300, 1e300};
  V3 cached_v{0, 0, 0};
  bool cached_valid = false;
  bool needs_update = true;
  bool degenerate = false;
};
const double EQUAL_VERTEX_THRESHOLD = 1e-12;  // VORONOI_DEFAULT_EQUAL_VERTEX_THRESHOLD, double-precision build

struct TriResult {
  V3 p;
  bool a, b, c;
};
TriResult closest_pt_triangle(V3 p, V3 a, V3 b, V3 c) {
  TriResult r{};
  V3 ab = b - a, ac = c - a, ap = p - a;
  double d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0 && d2 <= 0.0) return TriResult{a, true, false, false};
  V3 bp = p - b;
  double d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0 && d4 <= d3) return TriResult{b, false, true, false};
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
    double v = d1 / (d1 - d3);
    return TriResult{a + ab * v, true, true, false};
  }
  V3 cp = p - c;
  double d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0 && d5 <= d6) return TriResult{c, false, false, true};
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    double w = d2 / (d2 - d6);
    return TriResult{a + ac * w, true, false, true};
  }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    return TriResult{b + (c - b) * w, false, true, true};
  }
  double denom = 1.0 / (va + vb + vc);
  double v = vb * denom, w = vc * denom;
  r.p = a + ab * v + ac * w;
  r.a = r.b = r.c = true;
  return r;
}
// returns -1 degenerate, 0 inside, 1 outside
int point_outside_of_plane(V3 p, V3 a, V3 b, V3 c, V3 d) {
  V3 n = cross(b - a, c - a);
  double signp = dot(p - a, n), signd = dot(d - a, n);
  if (signd * signd < (1e-8 * 1e-8)) return -1;
  return signp * signd < 0.0 ? 1 : 0;
}
void simplex_remove(Simplex& s, int i) {
  s.n--;
  s.w[i] = s.w[s.n];
}
void simplex_reduce(Simplex& s, bool ua, bool ub, bool uc, bool ud) {
  if (s.n >= 4 && !ud) simplex_remove(s, 3);
  if (s.n >= 3 && !uc) simplex_remove(s, 2);
  if (s.n >= 2 && !ub) simplex_remove(s, 1);
  if (s.n >= 1 && !ua) simplex_remove(s, 0);
}
bool simplex_update(Simplex& s) {
  if (!s.needs_update) return s.cached_valid;
  s.needs_update = false;
  s.degenerate = false;
  const V3 o = v3(0, 0, 0);
  switch (s.n) {
    case 0:
      s.cached_valid = false;
      break;
    case 1:
      s.cached_v = s.w[0];
      s.cached_valid = true;
      break;
    case 2: {
      V3 from = s.w[0], to = s.w[1];
      V3 diff = o - from, v = to - from;
      double t = dot(v, diff);
      bool ua = true, ub = true;
      if (t > 0) {
        double dvv = dot(v, v);
        if (t < dvv) {
          t /= dvv;
        } else {
          t = 1;
          ua = false;
        }
      } else {
        t = 0;
        ub = false;
      }
      s.cached_v = from + v * t;
      simplex_reduce(s, ua, ub, false, false);
      s.cached_valid = true;
      break;
    }
    case 3: {
      TriResult r = closest_pt_triangle(o, s.w[0], s.w[1], s.w[2]);
      s.cached_v = r.p;
      simplex_reduce(s, r.a, r.b, r.c, false);
      s.cached_valid = true;
      break;
    }
    case 4: {
      V3 a = s.w[0], b = s.w[1], c = s.w[2], d = s.w[3];
      int oabc = point_outside_of_plane(o, a, b, c, d), oacd = point_outside_of_plane(o, a, c, d, b);
      int oadb = point_outside_of_plane(o, a, d, b, c), obdc = point_outside_of_plane(o, b, d, c, a);
      if (oabc < 0 || oacd < 0 || oadb < 0 || obdc < 0) {
        s.degenerate = true;
        s.cached_valid = false;
        break;
      }
      if (!oabc && !oacd && !oadb && !obdc) {  // origin inside the tetrahedron: the cores overlap
        s.cached_valid = true;
        s.cached_v = o;
        break;
      }
      double best = 1e300;
      V3 bp = o;
      bool ua = false, ub = false, uc = false, ud = false;
      if (oabc) {
        TriResult r = closest_pt_triangle(o, a, b, c);
        double sq = len2(r.p);
        if (sq < best) { best = sq; bp = r.p; ua = r.a; ub = r.b; uc = r.c; ud = false; }
      }
      if (oacd) {
        TriResult r = closest_pt_triangle(o, a, c, d);
        double sq = len2(r.p);
        if (sq < best) { best = sq; bp = r.p; ua = r.a; ub = false; uc = r.b; ud = r.c; }
      }
      if (oadb) {
        TriResult r = closest_pt_triangle(o, a, d, b);
        double sq = len2(r.p);
        if (sq < best) { best = sq; bp = r.p; ua = r.a; ub = r.c; uc = false; ud = r.b; }
      }
      if (obdc) {
        TriResult r = closest_pt_triangle(o, b, d, c);
        double sq = len2(r.p);
        if (sq < best) { best = sq; bp = r.p; ua = false; ub = r.a; uc = r.c; ud = r.b; }
      }
      s.cached_v = bp;
      simplex_reduce(s, ua, ub, uc, ud);
      s.cached_valid = true;
      break;
    }
  }
  return s.cached_valid;
}
bool simplex_in(const Simplex& s, V3 w) {
  for (int i = 0; i < s.n; i++)
    if (len2(s.w[i] - w) <= EQUAL_VERTEX_THRESHOLD) return true;
  if (w.x == s.last_w.x && w.y == s.last_w.y && w.z == s.last_w.z) return true;
  return false;
}

// ------------------------------------------------------------------------------------------------ penetration depth
// [BULLET] When the core shapes overlap, btGjkPairDetector hands over to btGjkEpaPenetrationDepthSolver (btGjkEpa2's
// expanding-polytope algorithm on the margin-inflated shapes) and p.getClosestPoints reports the NEGATIVE penetration
// depth as the contact distance -- what pyb_setup.py:452 stores in link_dist and ReachObs.compute_reward
// (reach.py:357-372) consumes on a terminal collision step.  Inflating both shapes by their margins adds
// margin_A + margin_B to the support function of the Minkowski difference in every direction, so
//     depth(inflated) = depth(cores) + margin_A + margin_B,   depth(cores) = min over unit n of h_{A-B}(n),
// and the oracle computes depth(cores) with an expanding polytope of its own, converged to EPA_TOL.  Bullet's EPA stops
// at EPA_ACCURACY = 1e-4 (btGjkEpa2.cpp) between its inner polytope and the support plane, i.e. its answer lies within
// 1e-4 m BELOW the value computed here; that envelope is the parity bar for penetrating queries and cannot be
// tightened without a pybullet to compare with [UNVERIFIED-BULLET].
struct EpaResult {
  double depth;    // >= 0: distance from the origin to the boundary of core_A - core_B
  int iterations;
  bool capped;
};
thread_local int g_last_epa_iterations = 0;  // probe for tests / sizing of the device workspace
const double EPA_TOL = 1.0e-9;
// The polytope lives in fixed slots exactly like the HIP path's LDS workspace (urgym_device.h epa_wave), so that both
// sides pick the same faces, in the same order, with the same arithmetic: at most EPA_MAX_VERTS points, hence at most
// 2 V - 4 = 92 <= EPA_MAX_FACES triangles.  Of 4889 random overlapping link <-> obstacle poses the median search took 16
// expansions, 99 % at most 38, the longest 76; stopping at 48 points (44 expansions) changes the depth by 1.9e-6 m at worst
// (one case above 1e-6, p99.9 5e-8) -- fifty times inside Bullet's own EPA accuracy of 1e-4 -- and takes 40 % off the longest
// searches, which is what a whole workgroup of the HIP path waits for.  A search that stops at the cap with more than
// EPA_CAP_RESIDUAL to gain is flagged URGYM_STATUS_GJK_ITER.
const int EPA_MAX_VERTS = 48, EPA_MAX_FACES = 128;
const double EPA_CAP_RESIDUAL = 1.0e-5;

// Works in B's frame like the device: X = pose of A in B's frame, w(n) = X S_A(X^T n) - S_B(-n).
EpaResult epa_core_depth(const Shape& A, const Shape& B) {
  X3 X;
  X.R = mul(M3{{{B.pose.R.m[0][0], B.pose.R.m[1][0], B.pose.R.m[2][0]}, {B.pose.R.m[0][1], B.pose.R.m[1][1], B.pose.R.m[2][1]},
               {B.pose.R.m[0][2], B.pose.R.m[1][2], B.pose.R.m[2][2]}}}, A.pose.R);
  X.t = mulT(B.pose.R, A.pose.t - B.pose.t);
  auto supp = [&](V3 n) { return (mul(X.R, support_local(A, mulT(X.R, n))) + X.t) - support_local(B, -n); };
  struct Face { int i, j, k; V3 n; double d; bool alive, degenerate; };
  V3 pts[EPA_MAX_VERTS];
  Face faces[EPA_MAX_FACES];
  for (auto& f : faces) f = Face{0, 0, 0, v3(0, 0, 0), 0.0, false, false};
  int nv = 0;
  auto make_face = [&](int i, int j, int k, V3 pi, V3 pj, V3 pk) {
    Face f{i, j, k, v3(0, 0, 0), 1e300, true, true};
    V3 n = cross(pj - pi, pk - pi);
    double l2 = len2(n);
    if (l2 > 1e-40) {
      f.n = n * (1.0 / std::sqrt(l2));
      f.d = dot(f.n, pi);
      f.degenerate = false;
    }
    return f;
  };
  // initial tetrahedron: support points of the four tetrahedral directions (it need not contain the origin yet: faces that
  // have the origin outside carry d < 0 and are expanded first)
  const double t = 0.5773502691896258;
  const V3 dirs[4] = {v3(t, t, t), v3(t, -t, -t), v3(-t, t, -t), v3(-t, -t, t)};
  for (int i = 0; i < 4; i++) pts[nv++] = supp(dirs[i]);
  const int tet[4][4] = {{0, 1, 2, 3}, {0, 3, 1, 2}, {0, 2, 3, 1}, {1, 3, 2, 0}};
  for (int f = 0; f < 4; f++) {
    int i = tet[f][0], j = tet[f][1], k = tet[f][2], o = tet[f][3];
    V3 n = cross(pts[j] - pts[i], pts[k] - pts[i]);
    if (dot(n, pts[o] - pts[i]) > 0) std::swap(j, k);  // outward
    faces[f] = make_face(i, j, k, pts[i], pts[j], pts[k]);
  }
  EpaResult res{0.0, 0, false};
  for (;;) {
    int best = -1;
    for (int f = 0; f < EPA_MAX_FACES; f++)
      if (faces[f].alive && (best < 0 || faces[f].d < faces[best].d)) best = f;  // ties: the lowest slot
    const Face bf = faces[best];
    const V3 w = supp(bf.n);
    const double gain = dot(bf.n, w) - bf.d;
    if (gain <= EPA_TOL || nv >= EPA_MAX_VERTS) {
      res.capped = gain > EPA_CAP_RESIDUAL;
      res.depth = bf.d > 0 ? bf.d : 0.0;  // d < 0: the origin is on (or a hair outside) the boundary -> cores just touch
      g_last_epa_iterations = res.iterations;
      return res;
    }
    res.iterations++;
    // faces that see w die; their directed edges are the rim candidates (ascending slot, edge order ij, jk, ki)
    int cand[3 * EPA_MAX_FACES], nc = 0;
    for (int f = 0; f < EPA_MAX_FACES; f++) {
      Face& F = faces[f];
      if (!F.alive) continue;
      if (F.degenerate || dot(F.n, w) - F.d > 1e-14) {
        F.alive = false;
        cand[nc++] = (F.i << 8) | F.j; cand[nc++] = (F.j << 8) | F.k; cand[nc++] = (F.k << 8) | F.i;
      }
    }
    // horizon = candidates whose reverse is not a candidate; new faces fill the free slots in ascending order
    int slot = 0;
    for (int c = 0; c < nc; c++) {
      const int rev = ((cand[c] & 255) << 8) | (cand[c] >> 8);
      bool found = false;
      for (int x = 0; x < nc; x++) found = found || cand[x] == rev;
      if (found) continue;
      while (slot < EPA_MAX_FACES && faces[slot].alive) slot++;
      if (slot >= EPA_MAX_FACES) {  // cannot happen while 2 V - 4 <= EPA_MAX_FACES; mirrors the device's exit
        res.capped = true;
        res.depth = bf.d > 0 ? bf.d : 0.0;
        g_last_epa_iterations = res.iterations;
        return res;
      }
      const int a = cand[c] >> 8, b = cand[c] & 255;
      faces[slot] = make_face(a, b, nv, pts[a], pts[b], w);
    }
    pts[nv++] = w;
  }
}

struct GjkResult {
  bool has_point;     // a closest point within max_dist was produced (pybullet getClosestPoints returns a non-empty list)
  double distance;    // signed distance between the margin-inflated shapes
  bool penetrating;   // cores overlap: the distance is -(penetration depth), see epa_core_depth
  int iterations;
};

// [BULLET] btGjkPairDetector::getClosestPointsNonVirtual restated for the double-precision build
// (REL_ERROR2 = 1e-12), as reached from p.getClosestPoints (pyb_setup.py:401,410,421,436,452) through
// btCollisionWorld::contactPairTest -> btCompoundCollisionAlgorithm -> btConvexConvexAlgorithm.
// distance = |closest(core_A - core_B)| - margin_A - margin_B; a point is reported when distance <= threshold.
// `start`: first separating axis (world frame, pointing from B to A); nullptr = Bullet's +Y (URGYM_GJK_START_BULLET).
GjkResult gjk_distance(const Shape& A, const Shape& B, double threshold, const V3* start = nullptr) {
  const double REL_ERROR2 = 1.0e-12;
  const double EPS = 2.220446049250313e-16;
  GjkResult res{false, 0.0, false, 0};
  // the pair detector works relative to the mid point of the two origins
  V3 offset = (A.pose.t + B.pose.t) * 0.5;
  Shape a = A, b = B;
  a.pose.t = a.pose.t - offset;
  b.pose.t = b.pose.t - offset;
  double margin = A.margin + B.margin;
  // btConvexConvexAlgorithm: maximumDistanceSquared = (marginA + marginB + breakingThreshold(0.02) + threshold)^2
  double max_d = margin + 0.02 + threshold;
  double max_d2 = max_d * max_d;
  V3 v = start ? *start : v3(0, 1, 0);
  Simplex s;
  double sq = 1e300;
  bool check_simplex = false;
  int degenerate = 0;
  int iter = 0;
  for (;;) {
    V3 p = support_world(a, -v), qw = support_world(b, v);
    V3 w = p - qw;
    double delta = dot(v, w);
    if (delta > 0 && delta * delta > sq * max_d2) {
      degenerate = 10;
      check_simplex = true;
      break;
    }
    if (simplex_in(s, w)) {
      degenerate = 1;
      check_simplex = true;
      break;
    }
    double f0 = sq - delta, f1 = sq * REL_ERROR2;
    if (f0 <= f1) {
      degenerate = f0 <= 0 ? 2 : 11;
      check_simplex = true;
      break;
    }
    s.last_w = w;
    s.w[s.n++] = w;
    s.needs_update = true;
    if (!simplex_update(s)) {
      degenerate = 3;
      check_simplex = true;
      break;
    }
    V3 nv = s.cached_v;
    if (len2(nv) < REL_ERROR2) {
      v = nv;
      degenerate = 6;
      check_simplex = true;
      break;
    }
    double prev = sq;
    sq = len2(nv);
    if (prev - sq <= EPS * prev) {
      check_simplex = true;
      degenerate = 12;
      break;
    }
    v = nv;
    if (iter++ > 1000) break;
    if (s.n == 4) {
      degenerate = 13;
      break;
    }
  }
  res.iterations = iter;
  bool valid = false;
  double distance = 0;
  if (check_simplex) {
    double l2 = len2(v);
    if (l2 < REL_ERROR2) degenerate = 5;
    if (l2 > EPS * EPS) {
      distance = std::sqrt(l2) - margin;
      valid = true;
    }
  }
  // Bullet enters its penetration-depth solver when the core shapes touch/overlap (checkPenetration && !isValid), and
  // also -- catchDegeneratePenetrationCase -- whenever the core distance is below gGjkEpaPenetrationTolerance = 0.001;
  // in that second case EPA's answer replaces GJK's only when it is deeper, and the two agree within EPA_ACCURACY, so
  // GJK's value stands here.
  if (!valid || degenerate == 5 || degenerate == 6 || degenerate == 13 || (degenerate == 3)) {
    double l2 = len2(v);
    if (!valid || l2 < REL_ERROR2) {
      EpaResult e = epa_core_depth(a, b);
      res.penetrating = true;
      res.has_point = true;
      res.distance = -(e.depth + margin);
      if (e.capped) res.iterations = 1001;  // reported as URGYM_STATUS_GJK_ITER
      return res;
    }
  }
  if (valid && (distance < 0 || distance * distance < max_d2)) {
    if (distance <= threshold) {  // MyContactResultCallback: cp.m_distance1 <= m_closestDistanceThreshold
      res.has_point = true;
      res.distance = distance;
    }
  }
  if (!res.has_point) res.distance = distance;  // still informative for callers that only test has_point
  return res;
}

// ------------------------------------------------------------------------------------------------ scene
// Static scene: table reach.py:615 + pyb_setup.py:802-811, track reach.py:616 + pyb_setup.py:835-844.
inline X3 pose_at(double x, double y, double z) { return X3{m3_identity(), v3(x, y, z)}; }
Shape scene_table() { return make_box(0.55, 0.9, 0.46, pose_at(0.5, 0.0, -0.12 - 0.46)); }
Shape scene_track() { return make_box(0.1, 0.55, 0.06, pose_at(0.0, 0.0, 0.0 - 0.06)); }
// obstacle: cylinder radius 0.05, height 0.4, local Z axis (reach.py:279-288, 626-635; pyb_setup.py:601-612)
Shape scene_obstacle(const X3& pose) { return make_cylinder_z(0.05, 0.4, pose); }

struct EnvView {  // pointers to one env's slots in the SoA buffers
  int n, N;
};

// URGYM_GJK_START_GUIDED (include/urgym.h): unit vector from `centre` to the mid point of the link's bounding capsule
// (data/ur5e_model.h UR5E_CAPSULE); +Y when the two coincide.  Not part of the reference: an opt-in search start.
V3 guided_axis(int l, const X3& pose, V3 centre) {
  const double* c = UR5E_CAPSULE[l - 1];
  V3 mid = mul(pose.R, v3(0.5 * (c[0] + c[3]), 0.5 * (c[1] + c[4]), 0.5 * (c[2] + c[5]))) + pose.t;
  V3 d = mid - centre;
  double n2 = dot(d, d);
  return n2 > 1e-12 ? d * (1.0 / std::sqrt(n2)) : v3(0, 1, 0);
}
V3 capsule_mid(int l, const X3& pose) {
  const double* c = UR5E_CAPSULE[l - 1];
  return mul(pose.R, v3(0.5 * (c[0] + c[3]), 0.5 * (c[1] + c[4]), 0.5 * (c[2] + c[5]))) + pose.t;
}

// PyBullet.get_link_distances (pyb_setup.py:439-456): links 2..6 vs obstacle, distance=5.0.
// scope == URGYM_LINK_DIST_WORKBENCH: per link the minimum over obstacle, table, track (include/urgym.h) -- the rule that
// reproduces the observations stored with the Sep-2023 Obs / Sta checkpoints (tests/test_reference_pins.py).
void link_distances(const X3 link[7], const Shape& obstacle, double out[5], int* status, int gjk_start, int scope) {
  const bool guided = gjk_start == URGYM_GJK_START_GUIDED;
  for (int i = 0; i < 5; i++) {
    V3 ax = guided_axis(i + 2, link[i + 2], obstacle.pose.t);
    GjkResult r = gjk_distance(make_hull(i + 2, link[i + 2]), obstacle, 5.0, guided ? &ax : nullptr);
    out[i] = r.distance;
    if (r.penetrating) *status |= URGYM_STATUS_PENETRATION;
    if (r.iterations > 1000) *status |= URGYM_STATUS_GJK_ITER;
    if (scope == URGYM_LINK_DIST_WORKBENCH) {
      Shape objs[2] = {scene_table(), scene_track()};
      for (int o = 0; o < 2; o++) {
        ax = guided_axis(i + 2, link[i + 2], objs[o].pose.t);
        GjkResult rb = gjk_distance(make_hull(i + 2, link[i + 2]), objs[o], 5.0, guided ? &ax : nullptr);
        if (rb.distance < out[i]) out[i] = rb.distance;
        if (rb.penetrating) *status |= URGYM_STATUS_PENETRATION;
        if (rb.iterations > 1000) *status |= URGYM_STATUS_GJK_ITER;
      }
    }
  }
}
// What-if switch (tools/closed_loop_ablation.py only): which groups of pairs check_collision tests -- bit 0 obstacle,
// bit 1 table + track, bit 2 self pairs.  7 = the reference.
int g_collision_groups = 7;
// PyBullet.check_collision (pyb_setup.py:382-429). has_obstacle mirrors `keys[5] == 'obstacle'` (398-399).
bool check_collision(const X3 link[7], bool has_obstacle, const Shape* obstacle, double margin, int gjk_start) {
  const bool guided = gjk_start == URGYM_GJK_START_GUIDED;
  V3 ax;
  if (has_obstacle && (g_collision_groups & 1))
    for (int l = 2; l <= 6; l++) {
      ax = guided_axis(l, link[l], obstacle->pose.t);
      if (gjk_distance(make_hull(l, link[l]), *obstacle, margin, guided ? &ax : nullptr).has_point) return true;
    }
  Shape objs[2] = {scene_table(), scene_track()};
  for (int o = 0; o < 2 && (g_collision_groups & 2); o++)
    for (int l = 2; l <= 6; l++) {
      ax = guided_axis(l, link[l], objs[o].pose.t);
      if (gjk_distance(make_hull(l, link[l]), objs[o], margin, guided ? &ax : nullptr).has_point) return true;
    }
  int start = 3;
  for (int la = 1; la < 4 && (g_collision_groups & 4); la++) {
    for (int lb = start; lb < 7; lb++) {
      ax = guided_axis(la, link[la], capsule_mid(lb, link[lb]));
      if (gjk_distance(make_hull(la, link[la]), make_hull(lb, link[lb]), margin, guided ? &ax : nullptr).has_point) return true;
    }
    start++;
  }
  return false;
}

// ------------------------------------------------------------------------------------------------ RNG
// Counter-based Philox4x32-10 (Salmon et al., SC'11).  The reference draws goal positions from a seeded numpy
// Generator and orientations from the unseeded global numpy RNG (utils.py:81-100), so its resets
// are not reproducible; the build defines its own stream, shared bit-for-bit by the oracle and the HIP path.
inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline double u01(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

struct Draws {
  double u[20];
};
Draws draw_attempt(uint64_t seed, uint32_t env, uint32_t episode, uint32_t attempt) {
  Draws d;
  for (uint32_t blk = 0; blk < 5; blk++) {
    uint32_t o[4];
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), env, episode, attempt, blk, o);
    for (int i = 0; i < 4; i++) d.u[blk * 4 + i] = u01(o[i]);
  }
  return d;
}
const double DEG = M_PI / 180.0;
// utils.sample_euler_constrained (utils.py:81-86): roll ~ U(-90,-180), pitch 0, yaw ~ U(0,-180) degrees
inline void sample_euler_constrained(double u_roll, double u_yaw, double rpy[3]) {
  rpy[0] = (-90.0 + (-180.0 - -90.0) * u_roll) * DEG;
  rpy[1] = 0.0 * DEG;
  rpy[2] = (0.0 + (-180.0 - 0.0) * u_yaw) * DEG;
}
// utils.sample_euler_obstacle (utils.py:88-100)
inline void sample_euler_obstacle(double u_sign, double u_roll, double u_pitch, double rpy[3]) {
  double roll = (u_sign < 0.5) ? (-30.0 + (-150.0 - -30.0) * u_roll) : (30.0 + (150.0 - 30.0) * u_roll);
  double pitch = (roll < -90.0 || roll > 90.0) ? (-30.0 + (-150.0 - -30.0) * u_pitch) : (30.0 + (150.0 - 30.0) * u_pitch);
  rpy[0] = roll * DEG;
  rpy[1] = pitch * DEG;
  rpy[2] = 0.0 * DEG;
}

// ------------------------------------------------------------------------------------------------ env
// ur5e.urdf:237,245,253,261,269,277: lower = -upper
const double UR5E_JOINT_LIMIT[6] = {2 * M_PI, 2 * M_PI, M_PI, 2 * M_PI, 2 * M_PI, 2 * M_PI};

struct Oracle {
  urgym_config cfg;
  urgym_buffers buf;
  uint64_t seed = 0;
  int obs_dim = 0, goal_dim = 0;
  char err[256] = {0};
};

inline double& S(double* base, int f, int n, int N) { return base[(size_t)f * N + n]; }

inline X3 obstacle_pose(const urgym_buffers& b, int n, int N) {
  Quat q{S(b.obst_quat, 0, n, N), S(b.obst_quat, 1, n, N), S(b.obst_quat, 2, n, N), S(b.obst_quat, 3, n, N)};
  return X3{quat_to_mat(q), v3(S(b.obst_pos, 0, n, N), S(b.obst_pos, 1, n, N), S(b.obst_pos, 2, n, N))};
}
inline void set_obstacle_pose(const urgym_buffers& b, int n, int N, const double pos_rpy[6]) {
  // PyBullet.set_base_pose (pyb_setup.py:305-317): euler -> getQuaternionFromEuler -> resetBasePositionAndOrientation
  Quat q = quat_from_euler_bullet(pos_rpy[3], pos_rpy[4], pos_rpy[5]);
  for (int i = 0; i < 3; i++) S(b.obst_pos, i, n, N) = pos_rpy[i];
  S(b.obst_quat, 0, n, N) = q.x; S(b.obst_quat, 1, n, N) = q.y; S(b.obst_quat, 2, n, N) = q.z; S(b.obst_quat, 3, n, N) = q.w;
}
// ReachDyn.set_velocity (reach.py:728-753): v = (end-start)/T, omega = axis*angle/T of the start->end rotation.
void dyn_velocity(const double start[6], const double end[6], double T, double vel[6]) {
  for (int i = 0; i < 3; i++) vel[i] = (end[i] - start[i]) / T;
  Quat qe = quat_from_euler_bullet(end[3], end[4], end[5]), qs = quat_from_euler_bullet(start[3], start[4], start[5]);
  Quat dq = quat_difference_bullet(qs, qe);
  V3 axis;
  double angle;
  axis_angle_bullet(dq, &axis, &angle);
  vel[3] = axis.x * angle / T; vel[4] = axis.y * angle / T; vel[5] = axis.z * angle / T;
}
// [BULLET] PyBullet.step (pyb_setup.py:52-55): 20 x stepSimulation at 1/500 s.  For the mass-0 (fixed-base btMultiBody)
// obstacle each sub-step is
//   (a) btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof: the base's spatial acceleration is zero, but
//       its world-frame read-out vdot = R^T (a_lin + w_local x v_local) keeps the transport term, and
//       applyDeltaVeeMultiDof adds h * (w x v) to the base's LINEAR velocity (w itself is unchanged);
//   (b) btMultiBody::stepPositionsMultiDof: p += h v; R <- exp([w] h) R (world-frame omega), normalised.
// resetBaseVelocity (reach.py:745 -> pyb_setup.py:340-349) restores the task's (v, w) before every env step, so the
// drift of v restarts from the task's value each time.  PINNED by the reference's own data: the two consecutive
// UR5DynReach-v1 observations stored in Trained_Models/Trained_Dyn/best_model.zip are reproduced to 3e-8 by this rule
// and are off by 5e-4 m without (a) (tests/test_reference_pins.py).
// Displacement of the obstacle base over one env step for the twist (v, w) that set_velocity re-applies every step: a
// constant per episode, kept in rows 6..8 of the obst_vel state so that the HIP step adds it instead of looping.
void step_displacement(const double vel[6], int substeps, double h, double dp[3]) {
  V3 v = v3(vel[0], vel[1], vel[2]);
  const V3 w = v3(vel[3], vel[4], vel[5]);
  double p[3] = {0, 0, 0};
  for (int k = 0; k < substeps; k++) {
    v = v + cross(w, v) * h;
    p[0] += h * v.x; p[1] += h * v.y; p[2] += h * v.z;
  }
  dp[0] = p[0]; dp[1] = p[1]; dp[2] = p[2];
}
void integrate_obstacle(const urgym_buffers& b, int n, int N, const double vel[6], int substeps, double h) {
  double p[3] = {S(b.obst_pos, 0, n, N), S(b.obst_pos, 1, n, N), S(b.obst_pos, 2, n, N)};
  Quat q{S(b.obst_quat, 0, n, N), S(b.obst_quat, 1, n, N), S(b.obst_quat, 2, n, N), S(b.obst_quat, 3, n, N)};
  V3 v = v3(vel[0], vel[1], vel[2]);
  const V3 w = v3(vel[3], vel[4], vel[5]);
  for (int k = 0; k < substeps; k++) {
    v = v + cross(w, v) * h;
    p[0] += h * v.x; p[1] += h * v.y; p[2] += h * v.z;
    double ang = std::sqrt(len2(w));
    if (ang * h > 0.5 * (M_PI * 0.5)) ang = 0.5 * (M_PI * 0.5) / h;  // ANGULAR_MOTION_THRESHOLD
    V3 ax;
    if (ang < 0.001)
      ax = w * (0.5 * h - (h * h * h) * 0.020833333333 * ang * ang);
    else
      ax = w * (std::sin(0.5 * ang * h) / ang);
    Quat dq{ax.x, ax.y, ax.z, std::cos(ang * h * 0.5)};
    q = qnormalize(qmul(dq, q));
  }
  for (int i = 0; i < 3; i++) S(b.obst_pos, i, n, N) = p[i];
  S(b.obst_quat, 0, n, N) = q.x; S(b.obst_quat, 1, n, N) = q.y; S(b.obst_quat, 2, n, N) = q.z; S(b.obst_quat, 3, n, N) = q.w;
}

// obst_vel state rows: 0..5 the episode's twist (v, w), 6..8 its displacement per env step (include/urgym.h)
void store_velocity(const urgym_config& c, const urgym_buffers& b, int n, const double vel[6]) {
  const int N = c.num_envs;
  double dp[3];
  step_displacement(vel, 20, c.dt / 20.0, dp);
  for (int i = 0; i < 6; i++) S(b.obst_vel, i, n, N) = vel[i];
  for (int i = 0; i < 3; i++) S(b.obst_vel, 6 + i, n, N) = dp[i];
}

struct Pose6 {
  double ee[6];       // ee xyz + rpy (double)
  float ee32[6];      // as cast by _get_obs (core.py:253-257)
};

// RobotTaskEnv._get_obs (core.py:252-261) + UR5Ori.get_obs (UR5.py:320-325) + Reach*.get_obs (reach.py:189,307,653-657)
void write_obs(const Oracle& o, int n, const X3 link[7], float* obs_row, float* ach_row, float* des_row, const double vel_obs[6]) {
  const urgym_buffers& b = o.buf;
  const int N = o.cfg.num_envs;
  double rpy[3];
  euler_from_quat_bullet(mat_to_quat(link[6].R), rpy);
  float* p = obs_row;
  *p++ = (float)link[6].t.x; *p++ = (float)link[6].t.y; *p++ = (float)link[6].t.z;
  *p++ = (float)rpy[0]; *p++ = (float)rpy[1]; *p++ = (float)rpy[2];
  for (int i = 0; i < 6; i++) *p++ = (float)S(b.q, i, n, N);
  if (o.cfg.env_kind == URGYM_ENV_ORI) {
    for (int i = 0; i < 6; i++) *p++ = (float)S(b.goal, i, n, N);
  } else if (o.cfg.env_kind == URGYM_ENV_OBS) {
    for (int i = 0; i < 3; i++) *p++ = (float)S(b.goal, i, n, N);
    for (int i = 0; i < 6; i++) *p++ = (float)S(b.obst_start, i, n, N);  // reach.py:308 echoes the sampled pose
    for (int i = 0; i < 5; i++) *p++ = (float)S(b.link_dist, i, n, N);
  } else {
    for (int i = 0; i < 6; i++) *p++ = (float)S(b.goal, i, n, N);
    for (int i = 0; i < 3; i++) *p++ = (float)S(b.obst_pos, i, n, N);  // reach.py:454-455 / 654-655 read back from Bullet
    Quat q{S(b.obst_quat, 0, n, N), S(b.obst_quat, 1, n, N), S(b.obst_quat, 2, n, N), S(b.obst_quat, 3, n, N)};
    double orpy[3];
    euler_from_quat_bullet(q, orpy);
    for (int i = 0; i < 3; i++) *p++ = (float)orpy[i];
    if (o.cfg.env_kind == URGYM_ENV_DYN)  // ReachSta.get_obs has no velocity slot (reach.py:457)
      for (int i = 0; i < 6; i++) *p++ = (float)vel_obs[i];
    for (int i = 0; i < 5; i++) *p++ = (float)S(b.link_dist, i, n, N);
  }
  for (int i = 0; i < o.goal_dim; i++) {
    ach_row[i] = obs_row[i];  // achieved_goal = ee xyz (+ rpy)  (reach.py:192-195, 310-311, 659-662)
    des_row[i] = (float)S(b.goal, i, n, N);
  }
}

// Reach*.is_success (reach.py:212-215, 348-350, 755-758) on the float32 achieved goal and the float64 goal
bool is_success(const Oracle& o, const float* ach32, const double goal[6], double* d_out, double* th_out) {
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < o.goal_dim; i++) a[i] = (double)ach32[i];
  double d = pos_distance(a, goal);
  *d_out = d;
  if (o.cfg.env_kind == URGYM_ENV_OBS) {
    *th_out = 0;
    return d < o.cfg.distance_threshold;
  }
  double th = angular_distance(a, goal);
  *th_out = th;
  return (d < o.cfg.distance_threshold) && (th < o.cfg.ori_threshold);
}

void reset_env(Oracle& o, int n) {
  const urgym_config& c = o.cfg;
  const urgym_buffers& b = o.buf;
  const int N = c.num_envs;
  int status = 0;
  // UR5Ori.reset -> set_joint_neutral (UR5.py:327-332, 262)
  for (int i = 0; i < 6; i++) S(b.q, i, n, N) = c.neutral_q[i];
  X3 link[7];
  double qn[6];
  for (int i = 0; i < 6; i++) qn[i] = c.neutral_q[i];
  forward_kinematics(qn, link);
  uint32_t episode = (uint32_t)b.episode_id[n];
  double goal[6] = {0, 0, 0, 0, 0, 0}, start[6] = {0, 0, 0, 0, 0, 0}, end[6] = {0, 0, 0, 0, 0, 0};
  int attempt = 0;
  for (;; attempt++) {
    Draws d = draw_attempt(o.seed, (uint32_t)n, episode, (uint32_t)attempt);
    // _sample_goal (reach.py:206-210, 337-340, 715-720)
    for (int i = 0; i < 3; i++) goal[i] = c.goal_low[i] + (c.goal_high[i] - c.goal_low[i]) * d.u[i];
    if (c.env_kind != URGYM_ENV_OBS) sample_euler_constrained(d.u[3], d.u[4], goal + 3);
    if (c.env_kind == URGYM_ENV_ORI) break;
    // _sample_obstacle (reach.py:342-346, 722-726)
    for (int i = 0; i < 3; i++) start[i] = c.obst_low[i] + (c.obst_high[i] - c.obst_low[i]) * d.u[5 + i];
    sample_euler_obstacle(d.u[8], d.u[9], d.u[10], start + 3);
    bool fail;
    if (c.env_kind == URGYM_ENV_STA) {
      // reach.py:467-472: target box half 0.025 (reach.py:416-424) at the goal pose vs the obstacle at its sampled pose
      X3 tp{quat_to_mat(quat_from_euler_bullet(goal[3], goal[4], goal[5])), v3(goal[0], goal[1], goal[2])};
      X3 op{quat_to_mat(quat_from_euler_bullet(start[3], start[4], start[5])), v3(start[0], start[1], start[2])};
      GjkResult r = gjk_distance(make_box(0.025, 0.025, 0.025, tp), scene_obstacle(op), 5.0);
      fail = r.distance < c.target_clearance;
    } else if (c.env_kind == URGYM_ENV_OBS) {
      // reach.py:316-322: sphere target r=0.02 (reach.py:270-277) vs obstacle
      X3 op{quat_to_mat(quat_from_euler_bullet(start[3], start[4], start[5])), v3(start[0], start[1], start[2])};
      GjkResult r = gjk_distance(make_sphere(0.02, pose_at(goal[0], goal[1], goal[2])), scene_obstacle(op), 5.0);
      fail = r.distance < c.target_clearance;
    } else {
      for (int i = 0; i < 3; i++) end[i] = c.obst_low[i] + (c.obst_high[i] - c.obst_low[i]) * d.u[11 + i];
      sample_euler_obstacle(d.u[14], d.u[15], d.u[16], end + 3);
      // reach.py:668-675: target box half 0.025 (reach.py:617-625) at the goal pose vs obstacle at its END pose
      X3 tp{quat_to_mat(quat_from_euler_bullet(goal[3], goal[4], goal[5])), v3(goal[0], goal[1], goal[2])};
      X3 op{quat_to_mat(quat_from_euler_bullet(end[3], end[4], end[5])), v3(end[0], end[1], end[2])};
      GjkResult r = gjk_distance(make_box(0.025, 0.025, 0.025, tp), scene_obstacle(op), 5.0);
      double travel = pos_distance(end, start);
      fail = (r.distance < c.target_clearance) || (travel < c.min_travel);
    }
    if (!fail) break;
    if (attempt + 1 >= c.max_reset_tries) {
      status |= URGYM_STATUS_RESET_EXHAUSTED;
      break;
    }
  }
  for (int i = 0; i < 6; i++) S(b.goal, i, n, N) = goal[i];
  bool coll = false;
  if (c.env_kind != URGYM_ENV_ORI) {
    for (int i = 0; i < 6; i++) {
      S(b.obst_start, i, n, N) = start[i];
      S(b.obst_end, i, n, N) = end[i];  // Sta: zeros = static obstacle (ReachSta.__init__, reach.py:406)
    }
    set_obstacle_pose(b, n, N, start);  // reach.py:319 / 678
    double vel[6] = {0, 0, 0, 0, 0, 0};
    if (c.env_kind == URGYM_ENV_DYN) dyn_velocity(start, end, c.dyn_time_duration, vel);
    store_velocity(c, b, n, vel);
    Shape obst = scene_obstacle(obstacle_pose(b, n, N));
    if (c.check_collision) coll = check_collision(link, true, &obst, c.collision_margin, c.gjk_start);  // reach.py:323 / 679
    double ld[5];
    link_distances(link, obst, ld, &status, c.gjk_start, c.link_dist_scope);  // reach.py:324-325 / 680-681
    for (int i = 0; i < 5; i++) S(b.link_dist, i, n, N) = ld[i];
    if (coll) status |= URGYM_STATUS_RESET_COLLISION;
  }
  b.collision[n] = coll ? 1 : 0;
  b.step_count[n] = 0;
  b.episode_id[n] = (int32_t)(episode + 1);
  // the velocity slot of the reset observation is the STALE ReachDyn.velocity of the previous step (reach.py:657;
  // reset() does not clear it).  It lives in the observation buffer itself.
  double vel_obs[6] = {0, 0, 0, 0, 0, 0};
  float* obs_row = b.observation + (size_t)n * o.obs_dim;
  if (c.env_kind == URGYM_ENV_DYN)
    for (int i = 0; i < 6; i++) vel_obs[i] = (double)obs_row[24 + i];
  float* ach = b.achieved_goal + (size_t)n * o.goal_dim;
  float* des = b.desired_goal + (size_t)n * o.goal_dim;
  write_obs(o, n, link, obs_row, ach, des, vel_obs);
  double d, th;
  b.is_success[n] = is_success(o, ach, goal, &d, &th) ? 1 : 0;  // core.py:272
  b.terminated[n] = 0;
  b.truncated[n] = 0;
  b.reward[n] = 0.f;
  if (status) b.status[n] |= status;
}

// Reach*.set_goal / set_goal_and_obstacle (reach.py:202-204, 328-335, 702-713) + model_test.get_obs (model_test.py:11-23)
void refresh_env(Oracle& o, int n) {
  const urgym_config& c = o.cfg;
  const urgym_buffers& b = o.buf;
  const int N = c.num_envs;
  int status = 0;
  double q[6];
  for (int i = 0; i < 6; i++) q[i] = S(b.q, i, n, N);
  X3 link[7];
  forward_kinematics(q, link);
  double goal[6];
  for (int i = 0; i < 6; i++) goal[i] = S(b.goal, i, n, N);
  if (c.env_kind != URGYM_ENV_ORI) {
    double start[6], end[6];
    for (int i = 0; i < 6; i++) { start[i] = S(b.obst_start, i, n, N); end[i] = S(b.obst_end, i, n, N); }
    set_obstacle_pose(b, n, N, start);
    double vel[6] = {0, 0, 0, 0, 0, 0};
    if (c.env_kind == URGYM_ENV_DYN) dyn_velocity(start, end, c.dyn_time_duration, vel);
    store_velocity(c, b, n, vel);
    Shape obst = scene_obstacle(obstacle_pose(b, n, N));
    bool coll = c.check_collision ? check_collision(link, true, &obst, c.collision_margin, c.gjk_start) : false;
    double ld[5];
    link_distances(link, obst, ld, &status, c.gjk_start, c.link_dist_scope);
    for (int i = 0; i < 5; i++) S(b.link_dist, i, n, N) = ld[i];
    b.collision[n] = coll ? 1 : 0;
  }
  double vel_obs[6] = {0, 0, 0, 0, 0, 0};
  float* obs_row = b.observation + (size_t)n * o.obs_dim;
  if (c.env_kind == URGYM_ENV_DYN)
    for (int i = 0; i < 6; i++) vel_obs[i] = (double)obs_row[24 + i];
  float* ach = b.achieved_goal + (size_t)n * o.goal_dim;
  float* des = b.desired_goal + (size_t)n * o.goal_dim;
  write_obs(o, n, link, obs_row, ach, des, vel_obs);
  double d, th;
  b.is_success[n] = is_success(o, ach, goal, &d, &th) ? 1 : 0;
  if (status) b.status[n] |= status;
}

// RobotTaskEnv.step (core.py:303-317) for one env, then TimeLimit (UR_gym/__init__.py:41).
void step_env(Oracle& o, int n, const float* action) {
  const urgym_config& c = o.cfg;
  const urgym_buffers& b = o.buf;
  const int N = c.num_envs;
  int status = 0;
  // 1. UR5Ori.set_action (UR5.py:273-279, 304-318): float32 clip, float32 * pi, float32 * 0.1, then float64 add
  double q[6];
  for (int i = 0; i < 6; i++) {
    float a = action[i];
    a = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);
    volatile float t1 = a * (float)M_PI;
    volatile float t2 = t1 * 0.1f;
    q[i] = S(b.q, i, n, N) + (double)t2;
    S(b.q, i, n, N) = q[i];
    // resetJointState does not clamp (pyb_setup.py:338); past the URDF limit (ur5e.urdf:237-277) Bullet's limit
    // constraint would act during stepSimulation, which this restatement does not model: flagged, not altered
    if (std::fabs(q[i]) > UR5E_JOINT_LIMIT[i]) status |= URGYM_STATUS_JOINT_LIMIT;
  }
  // 2. ReachDyn.set_velocity (core.py:305-306; reach.py:728-753)
  double vel_obs[6] = {0, 0, 0, 0, 0, 0};
  int step_num = b.step_count[n];
  if (c.env_kind == URGYM_ENV_DYN) {
    if (step_num < c.dyn_motion_steps) {
      double start[6], end[6];
      for (int i = 0; i < 6; i++) { start[i] = S(b.obst_start, i, n, N); end[i] = S(b.obst_end, i, n, N); }
      dyn_velocity(start, end, c.dyn_time_duration, vel_obs);
    }
    // 3. sim.step (core.py:309): obstacle base integrates the velocity just set
    integrate_obstacle(b, n, N, vel_obs, 20, c.dt / 20.0);
  } else if (c.env_kind == URGYM_ENV_STA) {
    // core.py:307-308: set_velocity only when obstacle_end is not all-zero; ReachSta.set_velocity (reach.py:518-541):
    // full start->end twist (time_duration = 1) while the obstacle is farther than 0.05 from its end position
    double start[6], end[6];
    bool moving = false;
    for (int i = 0; i < 6; i++) { start[i] = S(b.obst_start, i, n, N); end[i] = S(b.obst_end, i, n, N); moving = moving || end[i] != 0.0; }
    if (moving) {
      double dx = end[0] - S(b.obst_pos, 0, n, N), dy = end[1] - S(b.obst_pos, 1, n, N), dz = end[2] - S(b.obst_pos, 2, n, N);
      if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.05) dyn_velocity(start, end, 1.0, vel_obs);
      integrate_obstacle(b, n, N, vel_obs, 20, c.dt / 20.0);
    }
  }
  b.step_count[n] = step_num + 1;
  X3 link[7];
  forward_kinematics(q, link);
  // 4. collision = task.check_collision() (core.py:310)
  bool has_obst = c.env_kind != URGYM_ENV_ORI;
  Shape obst{};
  if (has_obst) obst = scene_obstacle(obstacle_pose(b, n, N));
  bool coll = c.check_collision ? check_collision(link, has_obst, has_obst ? &obst : nullptr, c.collision_margin, c.gjk_start) : false;
  // 5. observation (core.py:311) — link_dist still holds the value of the previous compute_reward/reset
  float* obs_row = b.observation + (size_t)n * o.obs_dim;
  float* ach = b.achieved_goal + (size_t)n * o.goal_dim;
  float* des = b.desired_goal + (size_t)n * o.goal_dim;
  write_obs(o, n, link, obs_row, ach, des, vel_obs);
  // 6./7. terminated, info (core.py:313-315)
  double goal[6];
  for (int i = 0; i < 6; i++) goal[i] = S(b.goal, i, n, N);
  double d, th;
  bool succ = is_success(o, ach, goal, &d, &th);
  bool terminated = succ || coll;
  bool info_success = terminated ? !coll : false;
  // 8. reward (core.py:316)
  double reward = 0.0;
  if (c.env_kind == URGYM_ENV_ORI) {
    // ReachOri.compute_reward (reach.py:221-236)
    reward += succ ? c.w_success : 0.0;
    reward += d * c.w_distance;
    reward += th * c.w_orientation;
    reward += coll ? c.w_collision : 0.0;
  } else if (c.env_kind == URGYM_ENV_OBS) {
    // ReachObs.compute_reward (reach.py:356-374)
    double ld[5];
    link_distances(link, obst, ld, &status, c.gjk_start, c.link_dist_scope);
    reward += succ ? c.w_success : 0.0;
    reward += coll ? c.w_collision : 0.0;
    reward += c.w_distance * d;
    double sum = 0.0;
    for (int i = 0; i < 5; i++) {
      double change = ld[i] - S(b.link_dist, i, n, N);
      sum += (ld[i] < c.near_threshold) ? c.w_link[i] * change : 0.0;
      S(b.link_dist, i, n, N) = ld[i];
    }
    reward += sum;
  } else {
    // ReachDyn.compute_reward (reach.py:764-785): early returns leave link_dist untouched
    if (coll) {
      reward = c.w_collision;
    } else if (succ) {
      reward = c.w_success;
    } else {
      reward += c.w_distance * d;
      reward += c.w_orientation * th;
      double ld[5];
      link_distances(link, obst, ld, &status, c.gjk_start, c.link_dist_scope);
      double sum = 0.0;
      for (int i = 0; i < 5; i++) {
        double change = ld[i] - S(b.link_dist, i, n, N);
        sum += (ld[i] < c.near_threshold) ? c.w_link[i] * change : 0.0;
        S(b.link_dist, i, n, N) = ld[i];
      }
      reward += sum;
    }
  }
  if (reward != reward) status |= URGYM_STATUS_NAN;
  b.reward[n] = (float)reward;
  b.terminated[n] = terminated ? 1 : 0;
  b.truncated[n] = (b.step_count[n] >= c.max_episode_steps) ? 1 : 0;
  b.is_success[n] = info_success ? 1 : 0;
  b.collision[n] = coll ? 1 : 0;
  if (status) b.status[n] |= status;
}

template <class F>
void parallel_for(int N, int threads, F f) {
  if (threads <= 1 || N < 2 * threads) {
    for (int i = 0; i < N; i++) f(i);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([=]() {
      int lo = (int)((int64_t)N * t / threads), hi = (int)((int64_t)N * (t + 1) / threads);
      for (int i = lo; i < hi; i++) f(i);
    });
  for (auto& th : pool) th.join();
}

}  // namespace

// ================================================================================================ C entry points
extern "C" {

int urgym_oracle_config_default(int env_kind, int num_envs, urgym_config* c) {
  if (!c || env_kind < 0 || env_kind > 3 || num_envs <= 0) return URGYM_ERR_ARG;
  std::memset(c, 0, sizeof(*c));
  c->env_kind = env_kind;
  c->num_envs = num_envs;
  c->max_episode_steps = 100;
  c->auto_reset = 1;
  c->check_collision = 1;
  c->max_reset_tries = 4096;
  c->gjk_start = URGYM_GJK_START_BULLET;
  c->link_dist_scope = URGYM_LINK_DIST_OBSTACLE;
  c->dyn_motion_steps = 25;
  c->action_scale = M_PI * 0.1;
  c->dt = 20.0 / 500.0;
  c->distance_threshold = 0.05;
  c->ori_threshold = 0.0873;
  c->w_collision = -500;
  c->w_success = 200;
  c->near_threshold = 0.2;
  c->collision_margin = 0.01;
  c->target_clearance = 0.1;
  c->min_travel = 1.0;
  c->dyn_time_duration = 2.0;
  const double neutral[6] = {0.0, -1.5708, 0.0, -1.5708, 0.0, 0.0};
  for (int i = 0; i < 6; i++) c->neutral_q[i] = neutral[i];
  if (env_kind == URGYM_ENV_ORI) {
    c->w_distance = -70; c->w_orientation = -30;
    const double gl[3] = {0.3, -0.5, 0.0}, gh[3] = {0.75, 0.5, 0.2};
    for (int i = 0; i < 3; i++) { c->goal_low[i] = gl[i]; c->goal_high[i] = gh[i]; }
  } else if (env_kind == URGYM_ENV_OBS) {
    c->w_distance = -100; c->w_orientation = 0;
    const double gl[3] = {0.3, -0.5, -0.1}, gh[3] = {0.75, 0.5, 0.2}, ol[3] = {0.5, -0.5, 0.25}, oh[3] = {1.0, 0.5, 0.55};
    for (int i = 0; i < 3; i++) { c->goal_low[i] = gl[i]; c->goal_high[i] = gh[i]; c->obst_low[i] = ol[i]; c->obst_high[i] = oh[i]; }
    for (int i = 0; i < 5; i++) c->w_link[i] = 100.0;
  } else {
    c->w_distance = -70; c->w_orientation = -30;
    const double gl[3] = {0.4, -0.5, 0.0}, gh[3] = {0.75, 0.5, 0.2}, ol[3] = {0.5, -0.8, 0.25}, oh[3] = {1.2, 0.8, 0.75};
    const double sgl[3] = {0.3, -0.5, 0.0}, sgh[3] = {0.75, 0.5, 0.2}, sol[3] = {0.5, -0.5, 0.25}, soh[3] = {1.0, 0.5, 0.55};  // reach.py:385-388
    const bool sta = env_kind == URGYM_ENV_STA;
    for (int i = 0; i < 3; i++) {
      c->goal_low[i] = sta ? sgl[i] : gl[i]; c->goal_high[i] = sta ? sgh[i] : gh[i];
      c->obst_low[i] = sta ? sol[i] : ol[i]; c->obst_high[i] = sta ? soh[i] : oh[i];
    }
    const double lw[5] = {8, 2.4, 1.2, 1.2, 0.2};
    double sum = 0;
    for (int i = 0; i < 5; i++) sum += lw[i];
    for (int i = 0; i < 5; i++) c->w_link[i] = lw[i] / sum * 50;
  }
  return URGYM_OK;
}

int urgym_oracle_create(const urgym_config* cfg, void** handle) {
  if (!cfg || !handle) return URGYM_ERR_ARG;
  Oracle* o = new Oracle();
  o->cfg = *cfg;
  o->obs_dim = cfg->env_kind == URGYM_ENV_ORI ? 18 : (cfg->env_kind == URGYM_ENV_OBS ? 26 : (cfg->env_kind == URGYM_ENV_STA ? 29 : 35));
  o->goal_dim = cfg->env_kind == URGYM_ENV_OBS ? 3 : 6;
  std::memset(&o->buf, 0, sizeof(o->buf));
  *handle = o;
  return URGYM_OK;
}
int urgym_oracle_destroy(void* h) {
  delete (Oracle*)h;
  return URGYM_OK;
}
int urgym_oracle_bind(void* h, const urgym_buffers* b) {
  if (!h || !b) return URGYM_ERR_ARG;
  ((Oracle*)h)->buf = *b;
  return URGYM_OK;
}
int urgym_oracle_reset(void* h, const uint8_t* mask, uint64_t seed, int threads) {
  Oracle* o = (Oracle*)h;
  if (!o) return URGYM_ERR_ARG;
  if (seed != UINT64_MAX) o->seed = seed;
  parallel_for(o->cfg.num_envs, threads, [=](int n) {
    if (!mask || mask[n]) reset_env(*o, n);
  });
  return URGYM_OK;
}
int urgym_oracle_refresh(void* h, const uint8_t* mask, int threads) {
  Oracle* o = (Oracle*)h;
  if (!o) return URGYM_ERR_ARG;
  parallel_for(o->cfg.num_envs, threads, [=](int n) {
    if (!mask || mask[n]) refresh_env(*o, n);
  });
  return URGYM_OK;
}
// counterpart of urgym_derive_obstacle_motion (include/urgym.h): rows 6..8 of obst_vel from the twist in rows 0..5
int urgym_oracle_derive_obstacle_motion(void* h) {
  Oracle* o = (Oracle*)h;
  if (!o) return URGYM_ERR_ARG;
  if (o->cfg.env_kind == URGYM_ENV_ORI) return URGYM_OK;
  const int N = o->cfg.num_envs;
  for (int n = 0; n < N; n++) {
    double vel[6];
    for (int i = 0; i < 6; i++) vel[i] = S(o->buf.obst_vel, i, n, N);
    store_velocity(o->cfg, o->buf, n, vel);
  }
  return URGYM_OK;
}
int urgym_oracle_step(void* h, const float* actions, int threads) {
  Oracle* o = (Oracle*)h;
  if (!o || !actions) return URGYM_ERR_ARG;
  const int od = o->obs_dim, gd = o->goal_dim;
  parallel_for(o->cfg.num_envs, threads, [=](int n) {
    step_env(*o, n, actions + (size_t)n * 6);
    const urgym_buffers& b = o->buf;
    if (o->cfg.auto_reset && (b.terminated[n] || b.truncated[n])) {
      // gymnasium VectorEnv autoreset: keep the terminal observation, then reset
      std::memcpy(b.final_observation + (size_t)n * od, b.observation + (size_t)n * od, sizeof(float) * od);
      std::memcpy(b.final_achieved_goal + (size_t)n * gd, b.achieved_goal + (size_t)n * gd, sizeof(float) * gd);
      std::memcpy(b.final_desired_goal + (size_t)n * gd, b.desired_goal + (size_t)n * gd, sizeof(float) * gd);
      float r = b.reward[n];
      uint8_t te = b.terminated[n], tr = b.truncated[n], su = b.is_success[n], co = b.collision[n];
      reset_env(*o, n);
      b.reward[n] = r; b.terminated[n] = te; b.truncated[n] = tr; b.is_success[n] = su; b.collision[n] = co;
    }
  });
  return URGYM_OK;
}

// ---- unit probes used by the parity tests -------------------------------------------------------------------
// link frames: out[7][12] = row-major 3x3 R then t, for PyBullet links 0..6
int urgym_oracle_fk(const double* q, double* out) {
  X3 link[7];
  forward_kinematics(q, link);
  for (int k = 0; k < 7; k++) {
    std::memcpy(out + k * 12, link[k].R.m, 9 * sizeof(double));
    out[k * 12 + 9] = link[k].t.x; out[k * 12 + 10] = link[k].t.y; out[k * 12 + 11] = link[k].t.z;
  }
  return URGYM_OK;
}
int urgym_oracle_ee_pose(const double* q, double* xyz_rpy) {
  X3 link[7];
  forward_kinematics(q, link);
  xyz_rpy[0] = link[6].t.x; xyz_rpy[1] = link[6].t.y; xyz_rpy[2] = link[6].t.z;
  euler_from_quat_bullet(mat_to_quat(link[6].R), xyz_rpy + 3);
  return URGYM_OK;
}
double urgym_oracle_distance(const double* a, const double* b) { return pos_distance(a, b); }
double urgym_oracle_angular_distance(const double* a6, const double* b6) { return angular_distance(a6, b6); }
void urgym_oracle_quat_from_euler(const double* rpy, double* q) {
  Quat r = quat_from_euler_bullet(rpy[0], rpy[1], rpy[2]);
  q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
}
void urgym_oracle_euler_from_quat(const double* q, double* rpy) { euler_from_quat_bullet(Quat{q[0], q[1], q[2], q[3]}, rpy); }
void urgym_oracle_dyn_velocity(const double* start, const double* end, double T, double* vel) { dyn_velocity(start, end, T, vel); }
// one env step (20 sub-steps of h = dt/20) of the obstacle base: pos_quat = xyz + quaternion xyzw, updated in place
void urgym_oracle_integrate_obstacle(double* pos_quat, const double* vel6, double dt) {
  double pos[3] = {pos_quat[0], pos_quat[1], pos_quat[2]}, quat[4] = {pos_quat[3], pos_quat[4], pos_quat[5], pos_quat[6]};
  urgym_buffers b{};
  b.obst_pos = pos;
  b.obst_quat = quat;
  integrate_obstacle(b, 0, 1, vel6, 20, dt / 20.0);
  for (int i = 0; i < 3; i++) pos_quat[i] = pos[i];
  for (int i = 0; i < 4; i++) pos_quat[3 + i] = quat[i];
}

static Shape probe_shape(int type, const double* params, const double* pose_xyz_quat) {
  Quat q{pose_xyz_quat[3], pose_xyz_quat[4], pose_xyz_quat[5], pose_xyz_quat[6]};
  X3 p{quat_to_mat(q), v3(pose_xyz_quat[0], pose_xyz_quat[1], pose_xyz_quat[2])};
  switch (type) {
    case SH_HULL: return make_hull((int)params[0], p);
    case SH_CYLZ: return make_cylinder_z(params[0], params[1], p);
    case SH_BOX: return make_box(params[0], params[1], params[2], p);
    default: return make_sphere(params[0], p);
  }
}
// generic closest-distance probe: type 0 hull(params[0]=link 1..6), 1 cylinderZ(radius,height), 2 box(half xyz), 3 sphere(r)
// pose = xyz + quaternion xyzw.  out = {has_point, distance, penetrating, iterations}
int urgym_oracle_closest(int type_a, const double* par_a, const double* pose_a, int type_b, const double* par_b,
                         const double* pose_b, double threshold, double* out) {
  GjkResult r = gjk_distance(probe_shape(type_a, par_a, pose_a), probe_shape(type_b, par_b, pose_b), threshold);
  out[0] = r.has_point; out[1] = r.distance; out[2] = r.penetrating; out[3] = r.iterations;
  return URGYM_OK;
}
// link distances + collision for a joint vector and an obstacle pose (xyz+quat); has_obstacle=0 -> Ori rules
int urgym_oracle_query(const double* q, const double* obst_pose, int has_obstacle, double margin, int gjk_start, int scope, double* ld5, int* collision) {
  X3 link[7];
  forward_kinematics(q, link);
  int status = 0;
  Shape obst{};
  if (has_obstacle) {
    Quat qq{obst_pose[3], obst_pose[4], obst_pose[5], obst_pose[6]};
    obst = scene_obstacle(X3{quat_to_mat(qq), v3(obst_pose[0], obst_pose[1], obst_pose[2])});
    link_distances(link, obst, ld5, &status, gjk_start, scope);
  }
  *collision = check_collision(link, has_obstacle != 0, has_obstacle ? &obst : nullptr, margin, gjk_start) ? 1 : 0;
  return status;
}
void urgym_oracle_set_emulation(int flags) { g_emulate = flags; }
void urgym_oracle_set_collision_groups(int bits) { g_collision_groups = bits; }
int urgym_oracle_last_epa_iterations(void) { return g_last_epa_iterations; }
void urgym_oracle_set_primitive_margin(double m) { g_prim_margin_override = m; }
void urgym_oracle_philox(uint64_t seed, uint32_t env, uint32_t episode, uint32_t attempt, double* u20) {
  Draws d = draw_attempt(seed, env, episode, attempt);
  std::memcpy(u20, d.u, sizeof(d.u));
}
int urgym_oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
