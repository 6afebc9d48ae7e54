// urgym_device.h — device-side math for the fused UR5e reach kernels (gfx950 only).
//
// Numeric plan: the reference evaluates this path in float64 (pybullet's double build, numpy,
// scipy) and casts to float32 only at the observation boundary, so the device does the same: FK chain, Euler /
// quaternion conversions, pose distances, the GJK simplex AND the hull support function are float64.  The support
// function stays cheap because an exact support map (candidate vertices per direction cell) replaces the scan over every vertex.
#pragma once
#include <stdint.h>
#if defined(URGYM_HOST_HARNESS)
// tests/device_harness.cpp compiles this very header with g++ to run the device algorithms on the CPU next to the
// oracle (debugging aid for parity work; never part of the product build)
#include <cmath>
#define __device__
#define __forceinline__ inline
#define URGYM_LDS
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
using std::fma; using std::sqrt; using std::fabs; using std::fmin; using std::fmax; using std::acos; using std::asin; using std::atan2;
#else
#include <hip/hip_runtime.h>
#define URGYM_LDS __attribute__((address_space(3)))
#endif

// diagnostic hook (-DURGYM_STAMPS build of urgym_hip.hip, tools/phase_stamps.py): a per-wave profile in LDS.  prof[0] = last mark,
// prof[1 + i] = cycles of section i (a section ends at its mark); from prof[1 + PROF_SECTIONS] on, as uint32 pairs, counter c =
// (times the wave executed the marked code, lanes that were active there summed over those executions): how full the wave was
// wherever it went.  Nothing of it is compiled into the product.
#if defined(URGYM_STAMPS) && !defined(URGYM_HOST_HARNESS)
constexpr int PROF_SECTIONS = 10, PROF_COUNTERS = 20, PROF_WORDS = 1 + PROF_SECTIONS + PROF_COUNTERS;
// counters: 0 loop trip (lanes = busy lanes), 1 first candidate record of a hull support call, 2 chained record, 3 segment case,
// 4 plane tests of the tetrahedron, 5 one face evaluation (triangle routine), 6..12 exits of the triangle routine (vertex A, B, edge
// AB, vertex C, edge AC, edge BC, face interior), 13 vertex reduction, 14 draw + set-up, 15 result handling of a finished query,
// 16 cylinder support, 17 box support, 18 early exits of the iteration (separating axis / duplicate / no progress)
#define URGYM_TRIP_MARK(i) trip_mark(r.clk, i)
#define URGYM_LANE_MARK(c) lane_mark(r.clk, c)
#define URGYM_LANE_MARK_AT(clk, c) lane_mark(clk, c)
#define URGYM_PROF_PARAM , URGYM_LDS unsigned long long* clk
#define URGYM_PROF_PASS(x) , x
__device__ __forceinline__ void trip_mark(URGYM_LDS unsigned long long* clk, int i) {
  if (clk == nullptr) return;  // a search outside the instrumented loop
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  clk[1 + i] += t - clk[0];
  clk[0] = t;
}
__device__ __forceinline__ void lane_mark(URGYM_LDS unsigned long long* clk, int c) {
  if (clk == nullptr) return;
  const unsigned long long m = __ballot(1);  // the lanes that execute this very code
  const int me = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  if (me == __builtin_ctzll(m)) {
    URGYM_LDS unsigned int* w = (URGYM_LDS unsigned int*)(clk + 1 + PROF_SECTIONS) + 2 * c;
    w[0] += 1u;
    w[1] += (unsigned int)__popcll(m);
  }
}
#else
#define URGYM_TRIP_MARK(i) do {} while (0)
#define URGYM_LANE_MARK(c) do {} while (0)
#define URGYM_LANE_MARK_AT(clk, c) do {} while (0)
#define URGYM_PROF_PARAM
#define URGYM_PROF_PASS(x)
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

// A rigid transform parked in LDS (per-lane slot, element k at p[k * stride]: r[0..8] then t.xyz).  The GJK keeps its
// pose operand there instead of in 24 VGPRs; it is re-read twice per iteration (18 + 12 ds_read_b64, conflict-free
// because consecutive lanes own consecutive addresses).
struct XRef {
  URGYM_LDS double* p;
  int stride;
  __device__ __forceinline__ double at(int k) const { return p[k * stride]; }
  __device__ __forceinline__ void set(int k, double v) const { p[k * stride] = v; }
};
__device__ __forceinline__ void store(XRef R, const X3& T) {
#pragma unroll
  for (int k = 0; k < 9; k++) R.set(k, T.r[k]);
  R.set(9, T.t.x); R.set(10, T.t.y); R.set(11, T.t.z);
}
__device__ __forceinline__ D3 rotT(XRef T, D3 v) {
  return d3(fma(T.at(0), v.x, fma(T.at(3), v.y, T.at(6) * v.z)), fma(T.at(1), v.x, fma(T.at(4), v.y, T.at(7) * v.z)),
            fma(T.at(2), v.x, fma(T.at(5), v.y, T.at(8) * v.z)));
}
__device__ __forceinline__ D3 apply(XRef T, D3 v) {
  return d3(fma(T.at(0), v.x, fma(T.at(1), v.y, T.at(2) * v.z)) + T.at(9), fma(T.at(3), v.x, fma(T.at(4), v.y, T.at(5) * v.z)) + T.at(10),
            fma(T.at(6), v.x, fma(T.at(7), v.y, T.at(8) * v.z)) + T.at(11));
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

// utils.distance (utils.py:5-31): L2 norm of the first three components
__device__ __forceinline__ double pos_distance(const double a[3], const double b[3]) {
  const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return sqrt(dx * dx + dy * dy + dz * dz);
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

// Convex-hull support tables in global memory (L2-resident, shared by every workgroup): an EXACT support map, no search.
// The support vertex of a convex polytope is piecewise constant in the direction: vertex p answers exactly the directions of its
// normal cone {d : d.p >= d.n for every neighbour n}, and the cones tile the sphere.  The directions are binned in a cube map
// (6 faces x DIRMAP_G x DIRMAP_G cells per hull); for every cell the host lists the CANDIDATES -- all vertices whose (slightly
// relaxed) cone meets the (slightly inflated) cell, found by clipping the cell's square against the cone's half-planes in the
// gnomonic plane of the face (urgym_tables_host.h).  The arg-max over a cell's candidates is therefore the arg-max over the whole
// hull, for every direction that maps to the cell -- the vertex the oracle's linear scan returns.  80 % of the cells have one
// candidate, 98.4 % at most two, 99.95 % at most four (one 128-byte record = one cache line); the few cells around the normal of
// a flat face shared by many vertices chain further records.  One support call = the cell's 2-byte code, then ONE record: two
// dependent memory round trips and four candidate evaluations, whatever the hull's size.
// (Rounds 1-2 climbed the hull's surface graph from a per-cell start vertex: 224-byte records of a vertex and seven neighbours,
//  chained for higher degrees, a round per climbing step and a last fetch of the winner's coordinates.  Measured per wave-wide GJK
//  iteration: 1.94 record rounds + 1.93 chained records at 23 / 4 active lanes -- 40 % of the iteration's time, a third of its vector
//  instructions, profiles/r3/lane_table_before.txt.)
struct alignas(16) D2 {
  double a, b;
};
struct alignas(128) CandRec {
  int next;        // next record of the same cell list, -1 = none
  int pad[3];
  D2 x[2], y[2], z[2];  // candidate j: (x[j/2], y[j/2], z[j/2]).{a|b}, by DESCENDING vertex id; unused slots repeat the last one
};
static_assert(sizeof(CandRec) == 128, "record layout");
#ifndef URGYM_DIRMAP_G
#define URGYM_DIRMAP_G 128
#endif
constexpr int DIRMAP_G = URGYM_DIRMAP_G;
constexpr int DIRMAP_CELLS = 6 * DIRMAP_G * DIRMAP_G;  // per hull
struct HullMap {
  const CandRec* __restrict__ recs;           // candidate records, shared by the cells with the same candidate set
  const unsigned short* __restrict__ cell;    // [6 hulls][6 faces][DIRMAP_G][DIRMAP_G] -> first record of the cell's list
};
// cell of a direction.  The cube map is the hardware's: V_CUBEID / V_CUBESC / V_CUBETC / V_CUBEMA_F32 (the texture unit's cube-map
// addressing, four instructions) give the face 0..5 = +x, -x, +y, -y, +z, -z, the two in-face coordinates and twice the major
// component; (sc, tc) / |ma| + 1/2 in [0, 1] then indexes the G x G cells of the face.  Face conventions (ISA manual; mirrored by
// cube_face() below, which the host uses to BUILD the table and the CPU harness to look it up):
//   major axis: z if |z| >= |x| and |z| >= |y|, else y if |y| >= |x|, else x
//   +x: sc = -z, tc = -y   -x: sc = z, tc = -y   +y: sc = x, tc = z   -y: sc = x, tc = -z   +z: sc = x, tc = -y   -z: sc = -x, tc = -y
// float32 on purpose: the host builds every cell's list for the cell inflated by far more than this arithmetic can err (margin 1e-5
// of the face's [-1, 1] square against ~1e-7), so a direction that lands in a neighbouring cell -- or, on an exact tie of two
// components, on the neighbouring FACE -- by rounding still finds its support vertex listed there.
struct CubeFace { int axis, au, av; double s, su, sv; };  // direction of in-face point (u, v): d[axis] = s, d[au] = su u, d[av] = sv v
#if defined(URGYM_HOST_HARNESS)
inline CubeFace cube_face(int face) {
#else
__host__ __device__ inline CubeFace cube_face(int face) {
#endif
  switch (face) {
    case 0: return CubeFace{0, 2, 1, 1.0, -1.0, -1.0};
    case 1: return CubeFace{0, 2, 1, -1.0, 1.0, -1.0};
    case 2: return CubeFace{1, 0, 2, 1.0, 1.0, 1.0};
    case 3: return CubeFace{1, 0, 2, -1.0, 1.0, -1.0};
    case 4: return CubeFace{2, 0, 1, 1.0, 1.0, -1.0};
    default: return CubeFace{2, 0, 1, -1.0, -1.0, -1.0};
  }
}
__device__ __forceinline__ int dirmap_cell(D3 d) {
  const float x = (float)d.x, y = (float)d.y, z = (float)d.z;
#ifdef URGYM_HOST_HARNESS
  const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
  float id, sc, tc, ma;
  if (az >= ax && az >= ay) { id = z < 0.0f ? 5.0f : 4.0f; sc = z < 0.0f ? -x : x; tc = -y; ma = 2.0f * z; }
  else if (ay >= ax) { id = y < 0.0f ? 3.0f : 2.0f; sc = x; tc = y < 0.0f ? -z : z; ma = 2.0f * y; }
  else { id = x < 0.0f ? 1.0f : 0.0f; sc = x < 0.0f ? z : -z; tc = -y; ma = 2.0f * x; }
  const float inv = 1.0f / fabsf(ma);
#else
  const float id = __builtin_amdgcn_cubeid(x, y, z), sc = __builtin_amdgcn_cubesc(x, y, z), tc = __builtin_amdgcn_cubetc(x, y, z);
  const float ma = __builtin_amdgcn_cubema(x, y, z);
  const float inv = __builtin_amdgcn_rcpf(fabsf(ma));  // (1 ulp: covered by the inflation of the cells)
#endif
  int iu = (int)((sc * inv + 0.5f) * (float)DIRMAP_G);   // (a zero direction gives NaN -> 0: any cell will do, nothing is searched with it)
  int iv = (int)((tc * inv + 0.5f) * (float)DIRMAP_G);
  iu = iu < 0 ? 0 : (iu > DIRMAP_G - 1 ? DIRMAP_G - 1 : iu);
  iv = iv < 0 ? 0 : (iv > DIRMAP_G - 1 ? DIRMAP_G - 1 : iv);
  return ((int)id * DIRMAP_G + iv) * DIRMAP_G + iu;
}

// d . v evaluated exactly like the oracle's scan ((x*dx + y*dy) + z*dz, no fused ops) so that near-tied vertices are
// ranked identically on both sides.
#if defined(URGYM_HOST_HARNESS)
__device__ __forceinline__ double vdot3(double x, double y, double z, D3 d) { return (x * d.x + y * d.y) + z * d.z; }
#else
// (explicitly rounded products and sums: this ranking must not be contracted into fused operations whatever the build's
//  -ffp-contract setting is, or near-tied vertices could be ranked differently from the oracle's scan)
__device__ __forceinline__ double vdot3(double x, double y, double z, D3 d) {
  return __dadd_rn(__dadd_rn(__dmul_rn(x, d.x), __dmul_rn(y, d.y)), __dmul_rn(z, d.z));
}
#endif

// branch-free "keep the better candidate".  ">=": a cell's candidates are listed by DESCENDING vertex id, so among exactly tied
// values the LOWEST id is the one kept -- the vertex the oracle's scan (first maximum) returns.  Near convergence the vertices of
// the closest face tie to the last bit now and then (11 of 292 000 support calls of a random census).
__device__ __forceinline__ void keep_better(double x, double y, double z, D3 d, double& best, D3& p) {
  const double t = vdot3(x, y, z, d);
  const bool g = t >= best;
  best = g ? t : best;
  p.x = g ? x : p.x; p.y = g ? y : p.y; p.z = g ? z : p.z;
}

// Support vertex of a hull in direction d (link frame), given the first record of d's cell (the cell's 2-byte code): the exact
// float64 arg-max over the cell's candidates.
__device__ __forceinline__ D3 hull_support_from(const HullMap& g, int rec, D3 d URGYM_PROF_PARAM) {
  double best = -1.0e300;
  D3 p = d3(0.0, 0.0, 0.0);
#if defined(URGYM_STAMPS) && !defined(URGYM_HOST_HARNESS)
  bool first = true;
#endif
  do {
#if defined(URGYM_STAMPS) && !defined(URGYM_HOST_HARNESS)
    URGYM_LANE_MARK_AT(clk, first ? 1 : 2);
    first = false;
#endif
    const CandRec& R = g.recs[rec];
    const int nextrec = R.next;
    const D2 x0 = R.x[0], x1 = R.x[1], y0 = R.y[0], y1 = R.y[1], z0 = R.z[0], z1 = R.z[1];  // the six 16-byte loads issue together
    keep_better(x0.a, y0.a, z0.a, d, best, p);
    keep_better(x0.b, y0.b, z0.b, d, best, p);
    keep_better(x1.a, y1.a, z1.a, d, best, p);
    keep_better(x1.b, y1.b, z1.b, d, best, p);
    rec = nextrec;
  } while (rec >= 0);
  return p;
}
__device__ __forceinline__ D3 hull_support(const HullMap& g, int h, D3 d URGYM_PROF_PARAM) {
  return hull_support_from(g, g.cell[h * DIRMAP_CELLS + dirmap_cell(d)], d URGYM_PROF_PASS(clk));
}

__device__ __forceinline__ D3 support_local(const HullMap& g, const ShapeDesc& s, D3 d URGYM_PROF_PARAM) {
  if (s.type == SH_HULL) {
    return hull_support(g, s.hull, d URGYM_PROF_PASS(clk));
  } else if (s.type == SH_CYLZ) {
    URGYM_LANE_MARK_AT(clk, 16);
    double sn = sqrt(d.x * d.x + d.y * d.y);
    double hz = d.z < 0.0 ? -s.hz : s.hz;
    if (sn != 0.0) {
      double k = s.hx / sn;
      return d3(d.x * k, d.y * k, hz);
    }
    return d3(s.hx, 0.0, hz);
  } else if (s.type == SH_BOX) {
    URGYM_LANE_MARK_AT(clk, 17);
    return d3(d.x >= 0.0 ? s.hx : -s.hx, d.y >= 0.0 ? s.hy : -s.hy, d.z >= 0.0 ? s.hz : -s.hz);
  }
  return d3(0.0, 0.0, 0.0);
}

// closest point of triangle (a,b,c) to the origin (Voronoi-region tests, Ericson RTCD §5.1.5); mask bit i = vertex i used
__device__ __forceinline__ D3 tri_closest(D3 a, D3 b, D3 c, int& mask URGYM_PROF_PARAM) {
  D3 ab = b - a, ac = c - a;
  double d1 = -dot(ab, a), d2 = -dot(ac, a);
  if (d1 <= 0.0 && d2 <= 0.0) { URGYM_LANE_MARK_AT(clk, 6); mask = 1; return a; }
  double d3_ = -dot(ab, b), d4 = -dot(ac, b);
  if (d3_ >= 0.0 && d4 <= d3_) { URGYM_LANE_MARK_AT(clk, 7); mask = 2; return b; }
  double vc = d1 * d4 - d3_ * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) { URGYM_LANE_MARK_AT(clk, 8); mask = 3; return a + ab * (d1 / (d1 - d3_)); }
  double d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0 && d5 <= d6) { URGYM_LANE_MARK_AT(clk, 9); mask = 4; return c; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { URGYM_LANE_MARK_AT(clk, 10); mask = 5; return a + ac * (d2 / (d2 - d6)); }
  double va = d3_ * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0) {
    URGYM_LANE_MARK_AT(clk, 11);
    mask = 6;
    return b + (c - b) * ((d4 - d3_) / ((d4 - d3_) + (d5 - d6)));
  }
  URGYM_LANE_MARK_AT(clk, 12);
  double den = 1.0 / (va + vb + vc);
  mask = 7;
  return a + ab * (vb * den) + ac * (vc * den);
}

enum { GJK_PENETRATING = 1, GJK_ITERCAP = 2, GJK_SEPARATED = 4, GJK_CLOSE = 8 /* stopped by the upper bound of a verdict query */ };

// Closest distance between the CORE shapes A (posed by T = pose of A in B's frame) and B (canonical frame):
// the same algorithm the reference reaches through p.getClosestPoints (pyb_setup.py:401-452), i.e. Bullet's
// btGjkPairDetector + btVoronoiSimplexSolver in the double-precision build — same start axis (world +Y, handed in
// as v0 in B's frame), same vertex-reduction order, same termination tests (relative 1e-12 on the squared distance,
// duplicate-vertex, no-progress, sliver tetrahedron) — so that the iterates, and with them the last bits of the
// distance, follow the oracle's.  Everything except the hull support scan is float64.
//   max_d : Bullet's early-out distance (marginA + marginB + 0.02 + query threshold) on the core distance; when a
//           separating axis proves the cores farther apart than that the search stops (GJK_SEPARATED).
// Returns the core distance |v|; GJK_PENETRATING when the cores touch/overlap (Bullet would enter EPA).
// Per-lane LDS slot of the GJK (element k at p[k * stride]):
//   0..11  pose of A in B's frame (row-major 3x3, then translation)
//   12..23 the simplex vertices w[0..3] (Minkowski-difference points)
// Keeping these 24 doubles out of the register file is what lets the float64 GJK run without scratch spills; the
// simplex is indexed dynamically (w[n] = ...) exactly like Bullet's arrays.
// (Bullet's inSimplex also tests w == m_lastW.  That test can never fire on its own: the vertex added last is still in
// the simplex unless the reduction dropped it, and then the closest point did not move, so the no-progress exit has
// already ended the loop in that same iteration.  m_lastW is therefore not stored.)
constexpr int GJK_SLOT_DOUBLES = 24;
__device__ __forceinline__ D3 ldw(XRef T, int i) { return d3(T.at(12 + 3 * i), T.at(13 + 3 * i), T.at(14 + 3 * i)); }
__device__ __forceinline__ void stw(XRef T, int i, D3 w) { T.set(12 + 3 * i, w.x); T.set(13 + 3 * i, w.y); T.set(14 + 3 * i, w.z); }

// The search is RESUMABLE (gjk_begin + one gjk_iterate per loop trip) so that a lane can finish one query and start
// another while its wave keeps iterating: the kernels use that to let lanes that are done early pick up queued work.
struct GjkRun {
  D3 v;             // Bullet's m_cachedSeparatingAxis
  double sq;        // squaredDistance
  int n;            // simplex size (vertices in the LDS slot)
  int iter;
  int info;         // GJK_* flags, valid when done
  double core;      // result (core distance), valid when done
  bool done;
#if defined(URGYM_STAMPS) && !defined(URGYM_HOST_HARNESS)
  URGYM_LDS unsigned long long* clk;  // diagnostic build: the wave's section clock
#endif
};
__device__ __forceinline__ void gjk_begin(GjkRun& r, D3 v0) {
  r.v = v0;
  r.sq = 1.0e300;
  r.n = 0;
  r.iter = 0;
  r.info = 0;
  r.core = 0.0;
  r.done = false;
#if defined(URGYM_STAMPS) && !defined(URGYM_HOST_HARNESS)
  r.clk = nullptr;
#endif
}
// exit of the search: what Bullet does after its loop (checkSimplex / degenerate cases).  separated: the exit was the early-out on a
// separating axis (Bullet's degenerate case 10)
__device__ __forceinline__ void gjk_finish(GjkRun& r, bool check_simplex, bool separated) {
  const double REL_ERROR2 = 1.0e-12;
  const double l2 = len2(r.v);
  r.done = true;
  if (!check_simplex || l2 < REL_ERROR2) {
    if (!(r.info & GJK_ITERCAP)) r.info |= GJK_PENETRATING;
    r.core = 0.0;
    return;
  }
  if (separated) r.info |= GJK_SEPARATED;
  r.core = sqrt(l2);
}
// one iteration of btGjkPairDetector's loop.  max_d: Bullet's early-out distance of this query (handed in per call rather than
// kept in the run: callers derive it from the query's kind, which costs less than two VGPRs across the loop)
// verdict_d (0 = off): for a query that only asks "are the cores closer than verdict_d?".  |v| is the distance of a point of A - B from
// the origin, i.e. an UPPER bound of the core distance: once it is <= verdict_d the answer is yes whatever the search would still
// find, and the search stops there (Bullet converges first and compares then -- same verdict).  The matching lower bound is the
// early-out above: such a caller hands in max_d = verdict_d.
__device__ __forceinline__ void gjk_iterate(GjkRun& r, const HullMap& g, const ShapeDesc& A, XRef T, const ShapeDesc& B, double max_d,
                                            double verdict_d = 0.0) {
  const double REL_ERROR2 = 1.0e-12;
  const double EPS = 2.220446049250313e-16;
  // the (up to three) vertices already in the simplex, requested together with the pose: their LDS round trip runs under the support
  // calls instead of after them (a slot at or beyond r.n holds stale data and is masked where it matters)
  const D3 W0 = ldw(T, 0), W1 = ldw(T, 1), W2 = ldw(T, 2);
  D3 w;
  {
    // A a hull: its cell code is requested first, the support of B (for the cylinder a square root and a division: a dependent chain
    // of its own) is evaluated while that load travels, the candidates of A's cell are fetched and ranked afterwards
    // the pose of A is read from LDS ONCE per iteration (direction transform and point transform use the same nine numbers): one LDS
    // round trip instead of two on the dependent chain; the twelve doubles stay in registers across the support call
    X3 TX;
#pragma unroll
    for (int k = 0; k < 9; k++) TX.r[k] = T.at(k);
    TX.t = d3(T.at(9), T.at(10), T.at(11));
    const D3 dA = rotT(TX, -r.v);
    int rec = 0;
    if (A.type == SH_HULL) rec = g.cell[A.hull * DIRMAP_CELLS + dirmap_cell(dA)];
    const D3 q = support_local(g, B, r.v URGYM_PROF_PASS(r.clk));
    URGYM_TRIP_MARK(2);
    const D3 sA = (A.type == SH_HULL) ? hull_support_from(g, rec, dA URGYM_PROF_PASS(r.clk)) : support_local(g, A, dA URGYM_PROF_PASS(r.clk));
    const D3 p = apply(TX, sA);
    URGYM_TRIP_MARK(1);
    w = p - q;
  }
  // Every exit of the iteration only RECORDS that and how the search ends (fin: bit 0 = ends, bit 1 = without the simplex check, i.e.
  // overlapping cores / iteration cap, bit 2 = on a separating axis); the bookkeeping of an ended search -- a square root among it --
  // runs once, at the bottom, for all lanes that end in this trip, instead of once per exit that some lane happens to take.
  int fin = 0;
  const double delta = dot(r.v, w);
  if (delta > 0.0 && delta * delta > r.sq * (max_d * max_d)) { URGYM_LANE_MARK(18); fin = 1 | 4; }
  if (fin == 0) {
    // (bitwise on purpose: three independent comparisons, no short-circuit branches)
    const bool in = (((int)(r.n > 0) & (int)(len2(W0 - w) <= 1e-12)) | ((int)(r.n > 1) & (int)(len2(W1 - w) <= 1e-12)) |
                     ((int)(r.n > 2) & (int)(len2(W2 - w) <= 1e-12))) != 0;
    const double f0 = r.sq - delta, f1 = r.sq * REL_ERROR2;
    if (in || f0 <= f1) { URGYM_LANE_MARK(18); fin = 1; }
  }
  if (fin == 0) {
    stw(T, r.n, w);
    int n = r.n + 1;
    URGYM_TRIP_MARK(2);
    // ---- closest point of the simplex to the origin + vertex reduction
    // (the vertices the closest point rests on are kept as ONE per-lane bit mask, bit i = vertex i: four booleans that are live across
    //  the face loop would be four lane masks in scalar registers, merged with three scalar instructions each per pass -- and the loop
    //  is short of scalar registers)
    D3 nv = d3(0, 0, 0);
    bool valid = true;
    int used = 15;
    bool reduce = true;
    if (n == 1) {
      nv = w;
      reduce = false;
    } else if (n == 2) {
      URGYM_LANE_MARK(3);
      D3 s0 = W0;
      D3 e = w - s0;
      double t = -dot(e, s0);
      used = 3;
      if (t > 0.0) {
        double ee = dot(e, e);
        if (t < ee) t /= ee;
        else { t = 1.0; used = 2; }
      } else {
        t = 0.0;
        used = 1;
      }
      nv = s0 + e * t;
    } else {
      // n == 3: the triangle itself.  n == 4: the faces in Bullet's order ABC|D, ACD|B, ADB|C, BDC|A, each only when the
      // origin lies on its outer side.  Two passes: (1) the four plane tests, which every lane with a tetrahedron needs,
      // leave a bit mask of the faces to evaluate; (2) each lane then works through ITS faces, lowest first (Bullet's order:
      // ties go to the earlier face).  The lanes of a wave rarely have more than two such faces, so pass 2 runs the (one)
      // triangle routine about twice per wave-wide iteration instead of four times; a lane with a triangle has one "face".
      const bool tetra = (n == 4);
      auto face_vertices = [](int f, int& ia, int& ib, int& ic, int& io) {
        ia = (f == 3) ? 1 : 0;
        ib = (f == 0) ? 1 : ((f == 1) ? 2 : 3);
        ic = (f == 0) ? 2 : ((f == 1) ? 3 : ((f == 2) ? 1 : 2));
        io = (f == 0) ? 3 : ((f == 1) ? 1 : ((f == 2) ? 2 : 0));
      };
      int todo = 1;  // triangle: just face 0 = (w0, w1, w2)
      bool degen = false;
      if (tetra) {
        // the four plane tests as straight-line code on the vertices read above: four independent chains the scheduler can
        // interleave, instead of four dependent rounds that each start with an LDS round trip (same expressions, same bits)
        todo = 0;
        URGYM_LANE_MARK(4);
        auto plane = [&](D3 a, D3 b, D3 c, D3 o, int bit) {
          const D3 nrm = cross(b - a, c - a);
          const double signp = -dot(a, nrm), signd = dot(o - a, nrm);
          if (signd * signd < (1.0e-8 * 1.0e-8)) degen = true;
          else if (signp * signd < 0.0) todo |= bit;
        };
        plane(W0, W1, W2, w, 1);   // ABC | D
        plane(W0, W2, w, W1, 2);   // ACD | B
        plane(W0, w, W1, W2, 4);   // ADB | C
        plane(W1, w, W2, W0, 8);   // BDC | A
      }
      URGYM_TRIP_MARK(6);
      double best = 1.0e300;
      used = -1;  // no face evaluated yet
#pragma unroll 1
      while (todo) {
        const int f = __builtin_ctz((unsigned)todo);
        todo &= todo - 1;
        int ia, ib, ic, io;
        face_vertices(f, ia, ib, ic, io);
        const D3 a = ldw(T, ia), b = ldw(T, ib), c = ldw(T, ic);
        int m3;
        URGYM_LANE_MARK(5);
        const D3 pt = tri_closest(a, b, c, m3 URGYM_PROF_PASS(r.clk));
        const double l = len2(pt);
        if (used < 0 || l < best) {
          best = l;
          nv = pt;
          used = ((m3 & 1) ? (1 << ia) : 0) | ((m3 & 2) ? (1 << ib) : 0) | ((m3 & 4) ? (1 << ic) : 0);
        }
      }
      if (degen) {
        valid = false;  // sliver tetrahedron: Bullet's closest() fails, the previous v stands
        reduce = false;
      } else if (used < 0) {
        nv = d3(0, 0, 0);  // origin inside the tetrahedron
        reduce = false;
      }
    }
    URGYM_TRIP_MARK(7);
    if (reduce) {
      URGYM_LANE_MARK(13);
      // btVoronoiSimplexSolver::reduceVertices: remove unused vertices from the back, removeVertex(i): w[i] = w[--n]
      if (n >= 4 && !(used & 8)) { n--; }
      if (n >= 3 && !(used & 4)) { n--; stw(T, 2, ldw(T, n)); }
      if (n >= 2 && !(used & 2)) { n--; stw(T, 1, ldw(T, n)); }
      if (n >= 1 && !(used & 1)) { n--; stw(T, 0, ldw(T, n)); }
    }
    r.n = n;
    if (!valid) fin = 1;  // sliver tetrahedron
    else {
      const double nsq = len2(nv);
      if (nsq < REL_ERROR2) { r.v = nv; fin = 1; }
      else if (nsq <= verdict_d * verdict_d) { r.v = nv; r.info |= GJK_CLOSE; fin = 1; }
      else {
        const double prev = r.sq;
        r.sq = nsq;
        if (prev - nsq <= EPS * prev) fin = 1;  // no progress: the previous v stands
        else {
          r.v = nv;
          if (r.iter++ > 1000) { r.info |= GJK_ITERCAP; fin = 3; }
          else if (n == 4) fin = 3;  // the origin is inside the tetrahedron
        }
      }
    }
  }  // (fin == 0: the simplex step)
  if (fin) gjk_finish(r, !(fin & 2), (fin & 4) != 0);
}


// ---------------------------------------------------------------------------------------------- penetration depth (EPA)
// When the cores overlap Bullet reports the NEGATIVE penetration depth of the margin-inflated shapes (btGjkEpa2 behind
// btGjkPairDetector; pyb_setup.py:452 stores it in link_dist).  depth(inflated) = depth(cores) + margin_A + margin_B, and
// depth(cores) = min over unit n of h_{A-B}(n) is found here by an expanding polytope -- the same slot-by-slot algorithm as
// the oracle's epa_core_depth, converged to 1e-9 (Bullet's own EPA stops at 1e-4: its answer lies within 1e-4 m below).
//
// ONE WAVE per query, all 64 lanes with identical arguments and wave-uniform control flow: the (rare) queries that need it
// are served after the GJK pool has drained, when the per-lane GJK slots of LDS are free.  The wave's 24 x 64 doubles
// M[r][c] (r = row of s_pose, c = the wave's column) are its workspace:
//   M[0..11][0]                       pose of A in B's frame (the XRef the supports read, every lane the same address)
//   M[3 (v / 63) + k][1 + v % 63]     coordinate k of polytope point v < EPA_MAX_VERTS            (rows 0..5)
//   rows 6, 7 (cols 1..63) as int32   face f: i | j << 8 | k << 16 | alive << 24 | degenerate << 25
//   rows 8..10 (cols 1..63) as uint16 rim candidates: the directed edges of the faces that see the new point
//   row 11 (cols 1..63) as uint8      free face slots in ascending order
//   M[12 + 4 (f / 64) + k][f % 64]    plane of face f: unit normal (k = 0..2) and offset d (k = 3)   (rows 12..23)
// Lane c owns the faces f = c and c + 64.
constexpr int EPA_MAX_VERTS = 48, EPA_MAX_FACES = 128;   // (the oracle's header comment has the census behind the cap)
constexpr int EPA_FACE_GROUPS = EPA_MAX_FACES / 64;       // faces per lane
constexpr double EPA_TOL = 1.0e-9, EPA_CAP_RESIDUAL = 1.0e-5;
#if !defined(URGYM_HOST_HARNESS)
struct EpaWs {
  URGYM_LDS double* base;  // &M[0][0]
  int stride;              // doubles between rows
  __device__ __forceinline__ URGYM_LDS double* at(int r, int c) const { return base + r * stride + c; }
  __device__ __forceinline__ D3 point(int v) const {
    const int r = 3 * (v / 63), c = 1 + v % 63;
    return d3(*at(r, c), *at(r + 1, c), *at(r + 2, c));
  }
  __device__ __forceinline__ void set_point(int v, D3 p) const {
    const int r = 3 * (v / 63), c = 1 + v % 63;
    *at(r, c) = p.x; *at(r + 1, c) = p.y; *at(r + 2, c) = p.z;
  }
  __device__ __forceinline__ URGYM_LDS int* face_word(int f) const {
    return f < 126 ? (URGYM_LDS int*)at(6, 1) + f : (URGYM_LDS int*)at(7, 1) + (f - 126);
  }
  __device__ __forceinline__ URGYM_LDS unsigned short* rim(int c) const { return (URGYM_LDS unsigned short*)at(8 + c / 252, 1) + c % 252; }
  __device__ __forceinline__ URGYM_LDS unsigned char* free_slot(int r) const { return (URGYM_LDS unsigned char*)at(11, 1) + r; }
  __device__ __forceinline__ URGYM_LDS double* plane(int f, int k) const { return at(12 + 4 * (f >> 6) + k, f & 63); }
};
__device__ __forceinline__ void epa_wave_sync() {  // LDS traffic of one wave is in order; this only stops the compiler
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// plane of the triangle (pi, pj, pk): returns false for a degenerate one (kept alive, replaced by the next expansion)
__device__ __forceinline__ bool epa_plane(D3 pi, D3 pj, D3 pk, D3& n, double& d) {
  const D3 c = cross(pj - pi, pk - pi);
  const double l2 = len2(c);
  if (l2 > 1e-40) {
    n = c * (1.0 / sqrt(l2));
    d = dot(n, pi);
    return true;
  }
  n = d3(0, 0, 0);
  d = 1e300;
  return false;
}
// T: pose of A in B's frame, stored at M[0..11][0] of `ws` (XRef{ws.base, ws.stride}).  Returns depth(cores) >= 0.
__device__ inline double epa_wave(const HullMap& g, const ShapeDesc& A, const ShapeDesc& B, EpaWs ws, int lane, bool& capped) {
  const XRef T{ws.base, ws.stride};
  auto supp = [&](D3 n) -> D3 { return apply(T, support_local(g, A, rotT(T, n) URGYM_PROF_PASS(nullptr))) - support_local(g, B, -n URGYM_PROF_PASS(nullptr)); };
  const double t = 0.5773502691896258;
  {
    const D3 p0 = supp(d3(t, t, t)), p1 = supp(d3(t, -t, -t)), p2 = supp(d3(-t, t, -t)), p3 = supp(d3(-t, -t, t));
    if (lane == 0) { ws.set_point(0, p0); ws.set_point(1, p1); ws.set_point(2, p2); ws.set_point(3, p3); }
#pragma unroll
    for (int gi = 0; gi < EPA_FACE_GROUPS; gi++) *ws.face_word(lane + 64 * gi) = 0;
    epa_wave_sync();
    if (lane < 4) {
      // faces of the tetrahedron 0123, turned outward
      int i = lane == 3 ? 1 : 0, j = lane == 0 ? 1 : (lane == 1 ? 3 : (lane == 2 ? 2 : 3)), k = lane == 0 ? 2 : (lane == 1 ? 1 : (lane == 2 ? 3 : 2));
      const int o = lane == 0 ? 3 : (lane == 1 ? 2 : (lane == 2 ? 1 : 0));
      const D3 pi = ws.point(i), pj = ws.point(j), pk = ws.point(k), po = ws.point(o);
      if (dot(cross(pj - pi, pk - pi), po - pi) > 0.0) { const int tmp = j; j = k; k = tmp; }
      D3 n;
      double d;
      const bool ok = epa_plane(ws.point(i), ws.point(j), ws.point(k), n, d);
      *ws.plane(lane, 0) = n.x; *ws.plane(lane, 1) = n.y; *ws.plane(lane, 2) = n.z; *ws.plane(lane, 3) = d;
      *ws.face_word(lane) = i | (j << 8) | (k << 16) | (1 << 24) | (ok ? 0 : (1 << 25));
    }
    epa_wave_sync();
  }
  int nv = 4;
  capped = false;
  double depth = 0.0;
#pragma unroll 1
  for (;;) {
    // (1) the face closest to the origin, ties to the lowest slot
    double bd = 1.7e308;
    int bf = 1 << 20;
#pragma unroll
    for (int gi = 0; gi < EPA_FACE_GROUPS; gi++) {
      const int f = lane + 64 * gi;
      const bool alive = ((*ws.face_word(f)) >> 24) & 1;
      const double d = *ws.plane(f, 3);
      if (alive && d < bd) { bd = d; bf = f; }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double od = __shfl_xor(bd, off);
      const int of = __shfl_xor(bf, off);
      if (od < bd || (od == bd && of < bf)) { bd = od; bf = of; }
    }
    bf = __builtin_amdgcn_readfirstlane(bf);
    const D3 nb = d3(*ws.plane(bf, 0), *ws.plane(bf, 1), *ws.plane(bf, 2));
    const double db = *ws.plane(bf, 3);
    // (2) expand along its normal
    const D3 w = supp(nb);
    const double gain = dot(nb, w) - db;
    if (gain <= EPA_TOL || nv >= EPA_MAX_VERTS) {
      capped = gain > EPA_CAP_RESIDUAL;
      depth = db > 0.0 ? db : 0.0;
      break;
    }
    // (3) faces that see w die; their directed edges become rim candidates in ascending slot order
    int nvis_before = 0, nc;
    {
      bool vis[EPA_FACE_GROUPS];
      int word[EPA_FACE_GROUPS];
#pragma unroll
      for (int gi = 0; gi < EPA_FACE_GROUPS; gi++) {
        const int f = lane + 64 * gi;
        word[gi] = *ws.face_word(f);
        const bool alive = (word[gi] >> 24) & 1, degen = (word[gi] >> 25) & 1;
        const D3 n = d3(*ws.plane(f, 0), *ws.plane(f, 1), *ws.plane(f, 2));
        vis[gi] = alive && (degen || dot(n, w) - *ws.plane(f, 3) > 1e-14);
      }
#pragma unroll
      for (int gi = 0; gi < EPA_FACE_GROUPS; gi++) {
        const unsigned long long m = __ballot(vis[gi]);
        if (vis[gi]) {
          const int rank = nvis_before + __popcll(m & ((1ull << lane) - 1ull));
          const int i = word[gi] & 255, j = (word[gi] >> 8) & 255, k = (word[gi] >> 16) & 255;
          *ws.rim(3 * rank + 0) = (unsigned short)((i << 8) | j);
          *ws.rim(3 * rank + 1) = (unsigned short)((j << 8) | k);
          *ws.rim(3 * rank + 2) = (unsigned short)((k << 8) | i);
          *ws.face_word(lane + 64 * gi) = 0;
        }
        nvis_before += __popcll(m);
      }
      nc = 3 * nvis_before;
    }
    epa_wave_sync();
    // (4) the free slots, ascending
    int nfree = 0;
#pragma unroll
    for (int gi = 0; gi < EPA_FACE_GROUPS; gi++) {
      const int f = lane + 64 * gi;
      const bool fr = !(((*ws.face_word(f)) >> 24) & 1);
      const unsigned long long m = __ballot(fr);
      if (fr) *ws.free_slot(nfree + __popcll(m & ((1ull << lane) - 1ull))) = (unsigned char)f;
      nfree += __popcll(m);
    }
    epa_wave_sync();
    // (5) horizon = candidates whose reverse is not a candidate; one new face per horizon edge, into the free slots in order
    int nh = 0;
    bool overflow = false;
#pragma unroll 1
    for (int c0 = 0; c0 < nc; c0 += 64) {
      const int c = c0 + lane;
      const bool valid = c < nc;
      const int e = valid ? (int)*ws.rim(c) : 0;
      const int rev = ((e & 255) << 8) | (e >> 8);
      bool found = false;
      if (nc <= 64) {
        // (the usual case: every candidate sits in a lane -- ask the lanes instead of reading the list back from LDS one dependent
        //  round trip at a time, which was half of an expansion's time)
#pragma unroll 1
        for (int x = 0; x < nc; x++) found = found || (__builtin_amdgcn_readlane(e, x) == rev);
      } else {
#pragma unroll 1
        for (int x = 0; x < nc; x++) found = found || ((int)*ws.rim(x) == rev);
      }
      const bool hor = valid && !found;
      const unsigned long long m = __ballot(hor);
      if (hor) {
        const int r = nh + __popcll(m & ((1ull << lane) - 1ull));
        if (r < nfree) {
          const int f = *ws.free_slot(r);
          const int a = e >> 8, b = e & 255;
          D3 n;
          double d;
          const bool ok = epa_plane(ws.point(a), ws.point(b), w, n, d);
          *ws.plane(f, 0) = n.x; *ws.plane(f, 1) = n.y; *ws.plane(f, 2) = n.z; *ws.plane(f, 3) = d;
          *ws.face_word(f) = a | (b << 8) | (nv << 16) | (1 << 24) | (ok ? 0 : (1 << 25));
        }
      }
      nh += __popcll(m);
    }
    overflow = nh > nfree;
    if (lane == 0) ws.set_point(nv, w);
    nv++;
    epa_wave_sync();
    if (overflow) { capped = true; depth = db > 0.0 ? db : 0.0; break; }
  }
  return depth;
}
#endif  // !URGYM_HOST_HARNESS

// convenience wrapper: run one query to the end
__device__ __forceinline__ double gjk_core_distance(const HullMap& g, const ShapeDesc& A, XRef T, const ShapeDesc& B,
                                                    D3 v0, double max_d, int& info, double verdict_d = 0.0) {
  GjkRun r;
  gjk_begin(r, v0);
  while (!r.done) gjk_iterate(r, g, A, T, B, max_d, verdict_d);
  info = r.info;
  return r.core;
}

}  // namespace urgym
