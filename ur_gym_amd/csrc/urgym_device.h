// urgym_device.h — device-side math for the fused UR5e reach kernels (gfx950 only).
//
// Numeric plan: the reference evaluates this path in float64 (pybullet's double build, numpy,
// scipy) and casts to float32 only at the observation boundary, so the device does the same: FK chain, Euler /
// quaternion conversions, pose distances, the GJK simplex AND the hull support search are float64.  The support
// search stays cheap because it walks the hull's surface graph instead of scanning every vertex.
#pragma once
#include <stdint.h>
#if defined(URGYM_HOST_HARNESS)
// tests/device_harness.cpp compiles this very header with g++ to run the device algorithms on the CPU next to the
// oracle (debugging aid for parity work; never part of the product build)
#include <cmath>
#define __device__
#define __forceinline__ inline
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
using std::fma; using std::sqrt; using std::fabs; using std::fmin; using std::fmax; using std::acos; using std::asin; using std::atan2;
#else
#include <hip/hip_runtime.h>
#endif

namespace urgym {

struct D3 {
  double x, y, z;
};
__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 operator-(D3 a) { return d3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(D3 a, D3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ D3 cross(D3 a, D3 b) {
  return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double len2(D3 a) { return dot(a, a); }
__device__ __forceinline__ D3 sel(bool c, D3 a, D3 b) { return d3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

// row-major 3x3 + translation; every index is a compile-time constant so the whole thing lives in VGPRs
struct X3 {
  double r[9];
  D3 t;
};
__device__ __forceinline__ D3 rot(const X3& T, D3 v) {
  return d3(fma(T.r[0], v.x, fma(T.r[1], v.y, T.r[2] * v.z)), fma(T.r[3], v.x, fma(T.r[4], v.y, T.r[5] * v.z)),
            fma(T.r[6], v.x, fma(T.r[7], v.y, T.r[8] * v.z)));
}
__device__ __forceinline__ D3 rotT(const X3& T, D3 v) {
  return d3(fma(T.r[0], v.x, fma(T.r[3], v.y, T.r[6] * v.z)), fma(T.r[1], v.x, fma(T.r[4], v.y, T.r[7] * v.z)),
            fma(T.r[2], v.x, fma(T.r[5], v.y, T.r[8] * v.z)));
}
__device__ __forceinline__ D3 apply(const X3& T, D3 v) { return rot(T, v) + T.t; }
// C = A^-1 * B for rigid transforms
__device__ __forceinline__ X3 rel(const X3& A, const X3& B) {
  X3 C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.r[i * 3 + j] = fma(A.r[i], B.r[j], fma(A.r[3 + i], B.r[3 + j], A.r[6 + i] * B.r[6 + j]));
  C.t = rotT(A, B.t - A.t);
  return C;
}

struct Q4 {
  double x, y, z, w;
};
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ void quat_to_rot(Q4 q, double r[9]) {
  double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  double s = 2.0 / d;
  double xs = q.x * s, ys = q.y * s, zs = q.z * s;
  double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  r[0] = 1.0 - (yy + zz); r[1] = xy - wz; r[2] = xz + wy;
  r[3] = xy + wz; r[4] = 1.0 - (xx + zz); r[5] = yz - wx;
  r[6] = xz - wy; r[7] = yz + wx; r[8] = 1.0 - (xx + yy);
}
// rotation matrix -> quaternion, largest-diagonal branch selection (what Bullet's getLinkState hands to
// getEulerFromQuaternion; pyb_setup.py:244-248)
__device__ __forceinline__ Q4 rot_to_quat(const double r[9]) {
  double tr = r[0] + r[4] + r[8];
  Q4 q;
  if (tr > 0.0) {
    double s = sqrt(tr + 1.0);
    q.w = 0.5 * s;
    s = 0.5 / s;
    q.x = (r[7] - r[5]) * s; q.y = (r[2] - r[6]) * s; q.z = (r[3] - r[1]) * s;
  } else if (r[0] >= r[4] && r[0] >= r[8]) {  // i = 0
    double s = sqrt(r[0] - r[4] - r[8] + 1.0);
    q.x = 0.5 * s;
    s = 0.5 / s;
    q.w = (r[7] - r[5]) * s; q.y = (r[3] + r[1]) * s; q.z = (r[6] + r[2]) * s;
  } else if (r[4] >= r[8]) {  // i = 1
    double s = sqrt(r[4] - r[8] - r[0] + 1.0);
    q.y = 0.5 * s;
    s = 0.5 / s;
    q.w = (r[2] - r[6]) * s; q.z = (r[7] + r[5]) * s; q.x = (r[1] + r[3]) * s;
  } else {  // i = 2
    double s = sqrt(r[8] - r[0] - r[4] + 1.0);
    q.z = 0.5 * s;
    s = 0.5 / s;
    q.w = (r[3] - r[1]) * s; q.x = (r[2] + r[6]) * s; q.y = (r[5] + r[7]) * s;
  }
  return q;
}
// pybullet getQuaternionFromEuler (pyb_setup.py:151-152): q = qz(yaw) qy(pitch) qx(roll)
__device__ __forceinline__ Q4 quat_from_rpy(double roll, double pitch, double yaw) {
  double sr, cr, sp, cp, sy, cy;
  sincos(0.5 * roll, &sr, &cr);
  sincos(0.5 * pitch, &sp, &cp);
  sincos(0.5 * yaw, &sy, &cy);
  return Q4{sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
}
// pybullet getEulerFromQuaternion (pyb_setup.py:190,248) incl. the |sin pitch| >= 0.99999 branches
__device__ __forceinline__ void rpy_from_quat(Q4 q, double& roll, double& pitch, double& yaw) {
  const double HALF_PI = 1.5707963267948966;
  double sarg = -2.0 * (q.x * q.z - q.w * q.y);
  if (sarg <= -0.99999) {
    roll = 0.0; pitch = -HALF_PI; yaw = 2.0 * atan2(q.x, -q.y);
  } else if (sarg >= 0.99999) {
    roll = 0.0; pitch = HALF_PI; yaw = 2.0 * atan2(-q.x, q.y);
  } else {
    double sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
    roll = atan2(2.0 * (q.y * q.z + q.w * q.x), squ - sqx - sqy + sqz);
    pitch = asin(sarg);
    yaw = atan2(2.0 * (q.x * q.y + q.w * q.z), squ + sqx - sqy - sqz);
  }
}
// scipy Rotation.from_euler('ZYX',[a0,a1,a2]) as utils.angular_distance uses it (utils.py:47-55): qz(a0) qy(a1) qx(a2)
__device__ __forceinline__ Q4 quat_ZYX(double a0, double a1, double a2) {
  double s0, c0, s1, c1, s2, c2;
  sincos(0.5 * a0, &s0, &c0);
  sincos(0.5 * a1, &s1, &c1);
  sincos(0.5 * a2, &s2, &c2);
  Q4 zy{-s0 * s1, c0 * s1, s0 * c1, c0 * c1};  // qz * qy
  return qmul(zy, Q4{s2, 0.0, 0.0, c2});
}
__device__ __forceinline__ double angular_distance(const double a[3], const double b[3]) {
  Q4 qa = quat_ZYX(a[0], a[1], a[2]), qb = quat_ZYX(b[0], b[1], b[2]);
  double d = qa.x * qb.x + qa.y * qb.y + qa.z * qb.z + qa.w * qb.w;
  d = fmin(1.0, fmax(-1.0, d));
  return 2.0 * acos(fabs(d));
}

// ---------------------------------------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u01(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

// ---------------------------------------------------------------------------------------------- shapes / GJK
enum { SH_HULL = 0, SH_CYLZ = 1, SH_BOX = 2, SH_POINT = 3 };
struct ShapeDesc {
  int type;
  int hull;           // SH_HULL: 0..5 = PyBullet link 1..6
  double hx, hy, hz;  // core half dims (cylinder: hx = core radius, hz = core half height)
};

// Convex-hull tables in global memory (L2-resident, < 1 MB, shared by every workgroup).
// The surface graph of each hull (Qhull triangulation, data/ur5e_model.h) is stored as NEIGHBOUR RECORDS: record i
// (i = global vertex id) holds the ids AND the exact float64 coordinates of up to 8 neighbours of vertex i, so one hill-
// climbing step is a single round trip of wide, independent loads (224 contiguous bytes) instead of the dependent
// chain offset -> ids -> coordinates.  Vertices with more than 8 neighbours chain further records through `next`;
// unused slots repeat the vertex itself (never "better", the comparison is strict).
struct NbrRec {
  int next;               // next record of the same vertex, -1 = none
  int pad[3];
  unsigned short id[8];
  double x[8], y[8], z[8];
};
static_assert(sizeof(NbrRec) == 224, "record layout");
constexpr int HULL_SEEDS = 16;
struct SeedRec {          // well-spread start vertices of one hull
  int id[HULL_SEEDS];
  double x[HULL_SEEDS], y[HULL_SEEDS], z[HULL_SEEDS];
};
struct HullGraph {
  const double* __restrict__ verts;   // [NV][3] exact link-frame vertices
  const NbrRec* __restrict__ recs;    // [NV + overflow]
  const SeedRec* __restrict__ seeds;  // [6]
};

// d . v evaluated exactly like the oracle's scan ((x*dx + y*dy) + z*dz, no fused ops) so that near-tied vertices are
// ranked identically on both sides.
__device__ __forceinline__ double vdot3(double x, double y, double z, D3 d) { return (x * d.x + y * d.y) + z * d.z; }

// Support vertex of hull `h` in direction d by steepest-ascent hill climbing on the hull's surface graph, in float64.
// On a convex polytope a vertex with no better neighbour is a global maximiser of the linear function, so this is
// the exact arg-max (tools/gen_model.py re-checks that against brute force when it builds the graph) at ~16-32 dot
// products instead of one per vertex.  `cur` carries the previous answer of this GJK run (warm start); -1 = none.
__device__ __forceinline__ D3 hull_support_climb(const HullGraph& g, int h, D3 d, int& cur) {
  double best;
  D3 pt;
  if (cur < 0) {
    const SeedRec& S = g.seeds[h];
    best = -1.0e300;
    pt = d3(0, 0, 0);
#pragma unroll
    for (int s = 0; s < HULL_SEEDS; s++) {
      const double t = vdot3(S.x[s], S.y[s], S.z[s], d);
      if (t > best) { best = t; cur = S.id[s]; pt = d3(S.x[s], S.y[s], S.z[s]); }
    }
  } else {
    const double* p = g.verts + 3 * cur;
    pt = d3(p[0], p[1], p[2]);
    best = vdot3(pt.x, pt.y, pt.z, d);
  }
  for (;;) {
    int nxt = cur;
    int rec = cur;
    do {
      const NbrRec& R = g.recs[rec];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const double t = vdot3(R.x[j], R.y[j], R.z[j], d);
        if (t > best) { best = t; nxt = R.id[j]; pt = d3(R.x[j], R.y[j], R.z[j]); }
      }
      rec = R.next;
    } while (rec >= 0);
    if (nxt == cur) break;
    cur = nxt;
  }
  return pt;
}

__device__ __forceinline__ D3 support_local(const HullGraph& g, const ShapeDesc& s, D3 d, int& cur) {
  if (s.type == SH_HULL) {
    return hull_support_climb(g, s.hull, d, cur);
  } else if (s.type == SH_CYLZ) {
    double sn = sqrt(d.x * d.x + d.y * d.y);
    double hz = d.z < 0.0 ? -s.hz : s.hz;
    if (sn != 0.0) {
      double k = s.hx / sn;
      return d3(d.x * k, d.y * k, hz);
    }
    return d3(s.hx, 0.0, hz);
  } else if (s.type == SH_BOX) {
    return d3(d.x >= 0.0 ? s.hx : -s.hx, d.y >= 0.0 ? s.hy : -s.hy, d.z >= 0.0 ? s.hz : -s.hz);
  }
  return d3(0.0, 0.0, 0.0);
}

// closest point of triangle (a,b,c) to the origin (Voronoi-region tests, Ericson RTCD §5.1.5); mask bit i = vertex i used
__device__ __forceinline__ D3 tri_closest(D3 a, D3 b, D3 c, int& mask) {
  D3 ab = b - a, ac = c - a;
  double d1 = -dot(ab, a), d2 = -dot(ac, a);
  if (d1 <= 0.0 && d2 <= 0.0) { mask = 1; return a; }
  double d3_ = -dot(ab, b), d4 = -dot(ac, b);
  if (d3_ >= 0.0 && d4 <= d3_) { mask = 2; return b; }
  double vc = d1 * d4 - d3_ * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) { mask = 3; return a + ab * (d1 / (d1 - d3_)); }
  double d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0 && d5 <= d6) { mask = 4; return c; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { mask = 5; return a + ac * (d2 / (d2 - d6)); }
  double va = d3_ * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0) {
    mask = 6;
    return b + (c - b) * ((d4 - d3_) / ((d4 - d3_) + (d5 - d6)));
  }
  double den = 1.0 / (va + vb + vc);
  mask = 7;
  return a + ab * (vb * den) + ac * (vc * den);
}

enum { GJK_PENETRATING = 1, GJK_ITERCAP = 2, GJK_SEPARATED = 4 };

// Closest distance between the CORE shapes A (posed by T = pose of A in B's frame) and B (canonical frame):
// the same algorithm the reference reaches through p.getClosestPoints (pyb_setup.py:401-452), i.e. Bullet's
// btGjkPairDetector + btVoronoiSimplexSolver in the double-precision build — same start axis (world +Y, handed in
// as v0 in B's frame), same vertex-reduction order, same termination tests (relative 1e-12 on the squared distance,
// duplicate-vertex, no-progress, sliver tetrahedron) — so that the iterates, and with them the last bits of the
// distance, follow the oracle's.  Everything except the hull support scan is float64.
//   max_d : Bullet's early-out distance (marginA + marginB + 0.02 + query threshold) on the core distance; when a
//           separating axis proves the cores farther apart than that the search stops (GJK_SEPARATED).
// Returns the core distance |v|; GJK_PENETRATING when the cores touch/overlap (Bullet would enter EPA).
__device__ __forceinline__ double gjk_core_distance(const HullGraph& g, const ShapeDesc& A, const X3& T, const ShapeDesc& B,
                                                    D3 v0, double max_d, int& info) {
  const double REL_ERROR2 = 1.0e-12;
  const double EPS = 2.220446049250313e-16;
  info = 0;
  D3 v = v0;
  double sq = 1.0e300;
  const double max_d2 = max_d * max_d;
  D3 s0 = d3(0, 0, 0), s1 = s0, s2 = s0, s3 = s0;
  D3 last_w = d3(1e300, 1e300, 1e300);
  int n = 0;
  bool check_simplex = false;
  int degenerate = 0;
  int iter = 0;
  int curA = -1, curB = -1;  // warm starts of the two hull searches
  for (;;) {
    D3 p = apply(T, support_local(g, A, rotT(T, -v), curA));
    D3 q = support_local(g, B, v, curB);
    D3 w = p - q;
    double delta = dot(v, w);
    if (delta > 0.0 && delta * delta > sq * max_d2) { degenerate = 10; check_simplex = true; break; }
    {
      bool in = (n > 0 && len2(s0 - w) <= 1e-12) || (n > 1 && len2(s1 - w) <= 1e-12) || (n > 2 && len2(s2 - w) <= 1e-12) ||
                (n > 3 && len2(s3 - w) <= 1e-12) || (w.x == last_w.x && w.y == last_w.y && w.z == last_w.z);
      if (in) { degenerate = 1; check_simplex = true; break; }
    }
    double f0 = sq - delta, f1 = sq * REL_ERROR2;
    if (f0 <= f1) { degenerate = (f0 <= 0.0) ? 2 : 11; check_simplex = true; break; }
    last_w = w;
    if (n == 0) s0 = w; else if (n == 1) s1 = w; else if (n == 2) s2 = w; else s3 = w;
    n++;
    // ---- closest point of the simplex to the origin + vertex reduction
    D3 nv = d3(0, 0, 0);
    bool valid = true;
    bool ua = true, ub = true, uc = true, ud = true;
    bool reduce = true;
    if (n == 1) {
      nv = s0;
      reduce = false;
    } else if (n == 2) {
      D3 e = s1 - s0;
      double t = -dot(e, s0);
      if (t > 0.0) {
        double ee = dot(e, e);
        if (t < ee) t /= ee;
        else { t = 1.0; ua = false; }
      } else {
        t = 0.0;
        ub = false;
      }
      nv = s0 + e * t;
      uc = ud = false;
    } else if (n == 3) {
      int m;
      nv = tri_closest(s0, s1, s2, m);
      ua = m & 1; ub = m & 2; uc = m & 4; ud = false;
    } else {
      // faces in Bullet's order: ABC|D, ACD|B, ADB|C, BDC|A
      double best = 1.0e300;
      bool any_out = false, degen = false;
      ua = ub = uc = ud = false;
#pragma unroll 1
      for (int f = 0; f < 4; f++) {
        D3 a = (f == 3) ? s1 : s0;
        D3 b = (f == 0) ? s1 : ((f == 1) ? s2 : s3);
        D3 c = (f == 0) ? s2 : ((f == 1) ? s3 : ((f == 2) ? s1 : s2));
        D3 o = (f == 0) ? s3 : ((f == 1) ? s1 : ((f == 2) ? s2 : s0));
        D3 nrm = cross(b - a, c - a);
        double signp = -dot(a, nrm), signd = dot(o - a, nrm);
        if (signd * signd < (1.0e-8 * 1.0e-8)) degen = true;
        else if (signp * signd < 0.0) {
          int m3;
          D3 pt = tri_closest(a, b, c, m3);
          double l = len2(pt);
          if (!any_out || l < best) {
            best = l;
            nv = pt;
            const bool ma = m3 & 1, mb = m3 & 2, mc = m3 & 4;
            ua = (f == 3) ? false : ma;
            ub = (f == 0) ? mb : ((f == 2) ? mc : ((f == 3) ? ma : false));
            uc = (f == 0) ? mc : ((f == 1) ? mb : ((f == 3) ? mc : false));
            ud = (f == 0) ? false : ((f == 1) ? mc : mb);
          }
          any_out = true;
        }
      }
      if (degen) {
        valid = false;  // sliver tetrahedron: Bullet's closest() fails, the previous v stands
        reduce = false;
      } else if (!any_out) {
        nv = d3(0, 0, 0);  // origin inside the tetrahedron
        reduce = false;
      }
    }
    if (reduce) {
      // btVoronoiSimplexSolver::reduceVertices: remove unused vertices from the back, removeVertex(i): w[i] = w[--n]
      if (n >= 4 && !ud) { n--; }
      if (n >= 3 && !uc) { n--; D3 l = (n == 2) ? s2 : s3; s2 = l; }
      if (n >= 2 && !ub) { n--; D3 l = (n == 1) ? s1 : ((n == 2) ? s2 : s3); s1 = l; }
      if (n >= 1 && !ua) { n--; D3 l = (n == 0) ? s0 : ((n == 1) ? s1 : ((n == 2) ? s2 : s3)); s0 = l; }
    }
    if (!valid) { degenerate = 3; check_simplex = true; break; }
    double nsq = len2(nv);
    if (nsq < REL_ERROR2) { v = nv; degenerate = 6; check_simplex = true; break; }
    double prev = sq;
    sq = nsq;
    if (prev - sq <= EPS * prev) { degenerate = 12; check_simplex = true; break; }
    v = nv;
    if (iter++ > 1000) { info |= GJK_ITERCAP; break; }
    if (n == 4) { degenerate = 13; break; }
  }
  double l2 = len2(v);
  if (!check_simplex || l2 < REL_ERROR2) {
    if (!(info & GJK_ITERCAP)) info |= GJK_PENETRATING;
    return 0.0;
  }
  if (degenerate == 10) info |= GJK_SEPARATED;
  return sqrt(l2);
}

}  // namespace urgym
