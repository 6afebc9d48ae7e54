// urgym_hip.hip — fused UR5e reach environment kernels for MI355X (gfx950) + the C-ABI of include/urgym.h.
//
// One body template env_body<KIND, MODE> (MODE = STEP | RESET | REFRESH | PREFETCH) behind env_kernel<KIND, MODE> (one mode per launch)
// and env_step_fused<KIND> (the steady-state step: STEP workgroups + the PREFETCH refill of the previous step); device math in
// urgym_device.h.
// One workgroup = 4 waves (256 lanes) serving E environments (a launch parameter urgym_create picks: up to 128 for STEP, so that
// N = 65536 is ONE round of resident workgroups; up to 64 for RESET / REFRESH, 32 for PREFETCH).
//
//   P1   one lane per env (STEP: on the LAST wave(s), while the other waves already run their queries): joint update
//        q += f32(f32(clip(a) * pi) * 0.1) (UR5.py:273-279), one float64 FK pass, world bounding capsules, conservative
//        culling of the 19 table / track / self pairs of check_collision (pyb_setup.py:382-429) -> a 19-bit mask per env in
//        LDS; the end-effector read-out (pyb_setup.py:221-253); the env's set-up cache (sin / cos of the joints, advanced
//        obstacle pose) in a global scratch.  STEP keeps NO joint / obstacle state in LDS: a lane re-derives them from global
//        memory (joint_of_step, obstacle_of_step) until P1 has published, and reads the cache afterwards.
//   pool the closest-distance work of the workgroup: 5 E obstacle "tickets" (exact distance hull(link) <-> cylinder,
//        pyb_setup.py:439-456; 15 E with URGYM_LINK_DIST_WORKBENCH: table and track too) + the set bits of the pair masks
//        (boolean "closer than the margin?" queries, which stop as soon as the upper or the lower bound of the running search
//        decides the verdict).  Every lane advances ITS query by one GJK iteration per loop trip through one inlined, resumable
//        GJK body (gjk_begin / gjk_iterate); idle lanes draw the next item together.
//   EPA  the (rare) obstacle queries whose cores overlap and whose distance is consumed: penetration depth by an expanding
//        polytope, one wave per query, faces spread over the lanes (pyb_setup.py:452 stores a negative contact distance);
//        served by waves that have left the pool while the others still iterate.
//   P4   one lane per env: pose distances, success / collision / reward (reach.py:221-236, 356-374, 764-785), lagged
//        link_dist, state write-back; a finished env is reset inline from its prefetched episode record or appended to the
//        done list; observation rows -> LDS -> coalesced stores.
//
// RESET consumes the device-side done list (no host round trip): Philox-keyed rejection sampling of reach.py:313-326 /
// 664-683 by the whole first wave, then the neutral-pose link distances.  PREFETCH is RESET with an episode record as its
// output; its workgroups ride in the NEXT step's launch (env_step_fused).  REFRESH = Reach*.set_goal[_and_obstacle].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <new>
#include <vector>

#include "../../include/urgym.h"
#include "../../data/ur5e_model.h"
#include "urgym_device.h"
#include "urgym_tables_host.h"

using namespace urgym;


namespace {

constexpr int GROUP = 64;       // env slots per wave-wide pass
#ifndef URGYM_MAX_ENVS
#define URGYM_MAX_ENVS 64
#endif
constexpr int PREFETCH_MAX_ENVS = 32;       // envs per PREFETCH workgroup at most
constexpr int MAX_ENVS = URGYM_MAX_ENVS;   // most envs one RESET / REFRESH workgroup serves (one wave of per-env lanes)
// A STEP workgroup may serve up to two waves' worth of envs: with its link distances parked in a global scratch array instead
// of LDS the per-env LDS footprint is 12 bytes, so the 53.7 KB that three resident workgroups allow are not exceeded, and
// N = 65536 fits ONE round of resident workgroups (E = 90) instead of two rounds of 46 with a ragged second one.
constexpr int STEP_MAX_ENVS = 128;
#ifndef URGYM_WAVES
#define URGYM_WAVES 4
#endif
constexpr int WAVES = URGYM_WAVES;  // waves per workgroup
constexpr int THREADS = GROUP * WAVES;
constexpr uint32_t NO_ITEM = 0xFFFFFFFFu;
#ifndef URGYM_REFILL_MIN
#define URGYM_REFILL_MIN 16
#endif
constexpr int REFILL_MIN = URGYM_REFILL_MIN;
// Resident STEP workgroups per CU the kernels are compiled for (launch bounds -> VGPR budget: 3 -> 168, 2 -> 256).  3 is the product;
// 2 exists to MEASURE the register-rich variant (profiles/r3/EXPERIMENTS.md), it is not shipped.
#ifndef URGYM_RESIDENT
#define URGYM_RESIDENT 3
#endif
constexpr int SC_FRAMES = 19, SC_FIELDS = SC_FRAMES + 72;  // rows of KParams::sc_scratch
// bits of the per-env culling mask: table vs links 2..6, track vs links 2..6, the nine self pairs
constexpr int PAIR_TABLE = 0, PAIR_TRACK = 5, PAIR_SELF = 10;
__host__ __device__ constexpr int self_pair_bit(int la, int lb) {  // (1,3)(1,4)(1,5)(1,6)(2,4)(2,5)(2,6)(3,5)(3,6)
  return PAIR_SELF + (la == 1 ? 0 : (la == 2 ? 4 : 7)) + (lb - (la + 2));
}

enum { MODE_STEP = 0, MODE_RESET = 1, MODE_REFRESH = 2, MODE_PREFETCH = 3 };
enum { Q_TABLE = 0, Q_TRACK = 1, Q_SELF = 2 };

// Bullet collision margins.  URDF convex meshes: 0.001, hull un-shrunk.  Every primitive made by p.createCollisionShape
// (pyb_setup.py:748-752) ends with shape->setMargin(0.001) (the physics server's default collision margin), which
// btCylinderShape / btBoxShape answer by shrinking their core by the same amount: cylinder core r 0.049, h/2 0.199.
// PINNED by the reference's own observations (tests/test_reference_pins.py; the constructors' "safe margin" that round 1
// used is off by up to 1.7e-3 m there).  Obs target: btSphereShape, margin = its radius around a point core.
constexpr double M_HULL = 0.001, M_PRIM = 0.001;
constexpr double CYL_R = 0.05, CYL_H = 0.4, M_CYL = M_PRIM;
constexpr double TABLE_CX = 0.5, TABLE_CY = 0.0, TABLE_CZ = -0.58, TABLE_HX = 0.55, TABLE_HY = 0.9, TABLE_HZ = 0.46;
constexpr double M_TABLE = M_PRIM;
constexpr double TRACK_CX = 0.0, TRACK_CY = 0.0, TRACK_CZ = -0.06, TRACK_HX = 0.1, TRACK_HY = 0.55, TRACK_HZ = 0.06;
constexpr double M_TRACK = M_PRIM;
constexpr double TARGET_BOX_H = 0.025, M_TARGET_BOX = M_PRIM, TARGET_SPHERE_R = 0.02;

struct DevTables {
  double joint_rot[6][9];
  double joint_xyz[6][3];
  double capsule[6][7];
};
__constant__ DevTables c_tab;

// Prefetched episode records (DESIGN.md "auto-reset off the critical path"): everything a reset produces is a pure function of
// (seed, env, episode id), so the records of the next two episodes of every env are kept ready in handle-owned memory:
// slot = episode & 1; fields per env (float64): goal 0..5, obstacle start 6..11, obstacle end 12..17, obstacle velocity 18..23,
// obstacle quaternion 24..27, neutral-pose link distances 28..32, displacement per env step 33..35; ints: the episode id the record is for (-1 = none), status.
constexpr int REC_FIELDS = 36;
enum { REC_GOAL = 0, REC_START = 6, REC_END = 12, REC_VEL = 18, REC_QUAT = 24, REC_LD = 28, REC_DP = 33 };

struct KParams {
  urgym_config cfg;
  urgym_buffers buf;
  HullMap graph;    // exact support map of the six link hulls (device global memory; urgym_device.h)
  int obs_dim, goal_dim;
  uint32_t seed_lo, seed_hi;
  int pp;           // which done_count slot this launch appends to (STEP) / consumes (RESET)
  int copy_final;   // RESET: 1 = auto-reset (keep the step's reward/flags, save the terminal observation)
  int envs;         // envs per workgroup (chosen per launch: see urgym_create / do_step / launch_mode)
  int big_blocks;   // STEP only: the first big_blocks workgroups serve `envs` envs each, the rest `envs_tail` (two-tier launch
  int envs_tail;    // geometry: the last round of workgroups is made of smaller, shorter ones; 0 = uniform)
  // prefetched episode records (null / 0 when the feature is off)
  double* rec_d;    // [2][REC_FIELDS][N]
  int32_t* rec_i;   // [2][2][N]: {episode id of the record, status flags of its sampling}
  int2* rlist;      // (env, episode) entries: STEP appends the slots it consumed (refilled under the next step), RESET the
                    // episodes after the one it started (refilled synchronously), PREFETCH reads its work from here
  int* rcount;      // number of entries in rlist
  int rcap;         // capacity of rlist
  double* ld_scratch;  // [5][N]: the link distances of the running step (STEP keeps them here, not in LDS)
  double* sc_scratch;  // [SC_FIELDS][N]: STEP's per-env set-up cache, written by the env's P1 lane and read by every later draw of the
                       // same workgroup: rows 2k / 2k+1 = sin / cos of joint k after the action, rows 12..18 = obstacle position +
                       // quaternion after this step's motion (the six float64 sincos of a full forward-kinematics pass were three
                       // quarters of a set-up); with check_collision, rows SC_FRAMES + 12 (link - 1) .. + 11 = the frame of each
                       // link from P1's culling pass (3 x 3 rotation, then position), so that a draw loads its operand frame
                       // instead of multiplying the chain up again.
  int* rzero;       // a list counter this launch arms (sets to 0) for a later launch, or null
  int* rzero2;      // a second one
  int fallback_on;  // STEP with prefetch: 1 = the RESET / PREFETCH fallback launches follow this step (records may be stale)
  int prefetch;     // 1: STEP resets finished envs inline from valid records; RESET files refill entries
  int sc_frames;    // 1: sc_scratch also carries the link frames (rows SC_FRAMES ..)
  int inline_ori;   // 1: STEP of UR5OriReach-v1 resets finished envs inline (its reset is one goal draw, reach.py:197-200): no RESET launch
  float neutral_ach[6];  // end-effector position + Euler angles of the neutral pose, float32 as _get_obs casts them (set at create)
};
__device__ __forceinline__ double& REC(const KParams& P, int slot, int f, int n) {
  return P.rec_d[((size_t)slot * REC_FIELDS + f) * P.cfg.num_envs + n];
}
__device__ __forceinline__ int32_t& RECI(const KParams& P, int slot, int j, int n) {
  return P.rec_i[((size_t)slot * 2 + j) * P.cfg.num_envs + n];
}

__device__ __forceinline__ double& SOA(double* base, int f, int n, int N) { return base[(size_t)f * N + n]; }

// order-preserving map double <-> int64 (an involution): lets LDS keep "the smallest distance so far" with one ds_min_i64
__device__ __forceinline__ long long sortable(double x) {
  const long long k = __double_as_longlong(x);
  return k ^ ((k >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}
__device__ __forceinline__ double unsortable(long long k) { return __longlong_as_double(k ^ ((k >> 63) & 0x7FFFFFFFFFFFFFFFLL)); }

__device__ __forceinline__ ShapeDesc hull_desc(int link /*1..6*/) {
  ShapeDesc s;
  s.type = SH_HULL; s.hull = link - 1; s.hx = s.hy = s.hz = 0.0;
  return s;
}
__device__ __forceinline__ ShapeDesc cyl_desc() {
  ShapeDesc s;
  s.type = SH_CYLZ; s.hull = 0;
  s.hx = s.hy = CYL_R - M_CYL; s.hz = 0.5 * CYL_H - M_CYL;
  return s;
}
__device__ __forceinline__ ShapeDesc box_desc(double hx, double hy, double hz, double margin) {
  ShapeDesc s;
  s.type = SH_BOX; s.hull = 0;
  s.hx = hx - margin; s.hy = hy - margin; s.hz = hz - margin;
  return s;
}
__device__ __forceinline__ ShapeDesc point_desc() {
  ShapeDesc s;
  s.type = SH_POINT; s.hull = 0; s.hx = s.hy = s.hz = 0.0;
  return s;
}

// one joint of the URDF chain: T <- T * [R_fix | xyz] * Rz(q)   (ur5e.urdf:232-279)
__device__ __forceinline__ void fk_joint(X3& T, int k, double s, double c) {  // s, c = sin / cos of joint k
  const double* F = c_tab.joint_rot[k];
  const double* o = c_tab.joint_xyz[k];
  T.t = T.t + rot(T, d3(o[0], o[1], o[2]));
  // G = F * Rz(q): columns 0,1 mix, column 2 unchanged
  double g[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    g[i * 3 + 0] = F[i * 3 + 0] * c + F[i * 3 + 1] * s;
    g[i * 3 + 1] = F[i * 3 + 1] * c - F[i * 3 + 0] * s;
    g[i * 3 + 2] = F[i * 3 + 2];
  }
  double r[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r[i * 3 + j] = fma(T.r[i * 3], g[j], fma(T.r[i * 3 + 1], g[3 + j], T.r[i * 3 + 2] * g[6 + j]));
#pragma unroll
  for (int i = 0; i < 9; i++) T.r[i] = r[i];
}
__device__ __forceinline__ X3 identity_x3() {
  X3 T;
  T.r[0] = 1; T.r[1] = 0; T.r[2] = 0; T.r[3] = 0; T.r[4] = 1; T.r[5] = 0; T.r[6] = 0; T.r[7] = 0; T.r[8] = 1;
  T.t = d3(0, 0, 0);
  return T;
}

// closest distance between two segments (Ericson RTCD §5.1.9)
__device__ __forceinline__ double segseg_dist(D3 p1, D3 q1, D3 p2, D3 q2) {
  D3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
  double a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
  double s, t;
  const double EPS = 1e-18;
  if (a <= EPS && e <= EPS) return sqrt(len2(r));
  if (a <= EPS) {
    s = 0.0;
    t = fmin(1.0, fmax(0.0, f / e));
  } else {
    double c = dot(d1, r);
    if (e <= EPS) {
      t = 0.0;
      s = fmin(1.0, fmax(0.0, -c / a));
    } else {
      double b = dot(d1, d2), den = a * e - b * b;
      s = den > 1e-30 ? fmin(1.0, fmax(0.0, (b * f - c * e) / den)) : 0.0;
      t = (b * s + f) / e;
      if (t < 0.0) { t = 0.0; s = fmin(1.0, fmax(0.0, -c / a)); }
      else if (t > 1.0) { t = 1.0; s = fmin(1.0, fmax(0.0, (b - c) / a)); }
    }
  }
  D3 c1 = p1 + d1 * s, c2 = p2 + d2 * t;
  return sqrt(len2(c1 - c2));
}
// lower bound of the distance between segment (p,q) and an axis-aligned box (centre c, half h)
__device__ __forceinline__ double seg_box_lower_bound(D3 p, D3 q, double cx, double cy, double cz, double hx, double hy, double hz) {
  double gx = fmax(fmax((cx - hx) - fmax(p.x, q.x), fmin(p.x, q.x) - (cx + hx)), 0.0);
  double gy = fmax(fmax((cy - hy) - fmax(p.y, q.y), fmin(p.y, q.y) - (cy + hy)), 0.0);
  double gz = fmax(fmax((cz - hz) - fmax(p.z, q.z), fmin(p.z, q.z) - (cz + hz)), 0.0);
  return sqrt(gx * gx + gy * gy + gz * gz);
}

// ReachDyn.set_velocity (reach.py:728-753): v = (end-start)/T; omega = axis*angle/T of dq = nearest(q_end)*q_start^-1
__device__ void dyn_velocity(const double start[6], const double end[6], double T, double vel[6]) {
  for (int i = 0; i < 3; i++) vel[i] = (end[i] - start[i]) / T;
  Q4 q0 = quat_from_rpy(start[3], start[4], start[5]), q1 = quat_from_rpy(end[3], end[4], end[5]);
  double dm = (q0.x - q1.x) * (q0.x - q1.x) + (q0.y - q1.y) * (q0.y - q1.y) + (q0.z - q1.z) * (q0.z - q1.z) + (q0.w - q1.w) * (q0.w - q1.w);
  double dp = (q0.x + q1.x) * (q0.x + q1.x) + (q0.y + q1.y) * (q0.y + q1.y) + (q0.z + q1.z) * (q0.z + q1.z) + (q0.w + q1.w) * (q0.w + q1.w);
  if (!(dm < dp)) { q1.x = -q1.x; q1.y = -q1.y; q1.z = -q1.z; q1.w = -q1.w; }
  Q4 dq = qmul(q1, Q4{-q0.x, -q0.y, -q0.z, q0.w});
  double w = fmin(1.0, fmax(-1.0, dq.w));
  double angle = 2.0 * acos(w);
  double s2 = 1.0 - dq.w * dq.w;
  D3 axis = d3(1, 0, 0);
  if (!(s2 < 10.0 * 2.220446049250313e-16)) {
    double s = 1.0 / sqrt(s2);
    axis = d3(dq.x * s, dq.y * s, dq.z * s);
  }
  vel[3] = axis.x * angle / T; vel[4] = axis.y * angle / T; vel[5] = axis.z * angle / T;
}

// Displacement of the obstacle base over one env step = 20 Bullet sub-steps of h = dt / 20 (pyb_setup.py:52-55).  In every
// sub-step btMultiBody first adds h * (w x v) to the base's linear velocity (the transport term of the world-frame read-out
// of its zero spatial acceleration), then moves the base by h * v; resetBaseVelocity restores the task's twist before each
// env step (reach.py:745).  PINNED by the reference's two consecutive UR5DynReach-v1 observations (tests/test_reference_pins.py).
// The twist is constant over an episode, so is this vector: it is evaluated at reset / refresh and kept in rows 6..8 of obst_vel.
__device__ __forceinline__ void step_displacement(const double vel[6], double dt, double dp[3]) {
  const double h = dt / 20.0;
  D3 v = d3(vel[0], vel[1], vel[2]);
  const D3 w = d3(vel[3], vel[4], vel[5]);
  double px = 0.0, py = 0.0, pz = 0.0;
#pragma unroll 1
  for (int k = 0; k < 20; k++) {
    v = v + cross(w, v) * h;
    px += h * v.x; py += h * v.y; pz += h * v.z;
  }
  dp[0] = px; dp[1] = py; dp[2] = pz;
}
// one env step of the obstacle base: p += dp (above), R <- exp([w] dt) R  (20 sub-step rotations about a constant axis
// compose to a single exponential)
__device__ __forceinline__ void integrate_obstacle(double pos[3], Q4& q, const double vel[6], const double dp[3], double dt) {
  pos[0] += dp[0]; pos[1] += dp[1]; pos[2] += dp[2];
  D3 w = d3(vel[3], vel[4], vel[5]);
  double ang = sqrt(len2(w));
  if (ang > 0.0) {
    double s, c;
    sincos(0.5 * ang * dt, &s, &c);
    double k = s / ang;
    Q4 dq{w.x * k, w.y * k, w.z * k, c};
    Q4 r = qmul(dq, q);
    double nrm = 1.0 / sqrt(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
    q = Q4{r.x * nrm, r.y * nrm, r.z * nrm, r.w * nrm};
  }
}

// ---- RESET: sampling of a new episode (reach.py:197-200, 313-326, 664-683; samplers utils.py:81-100).
// Every test of a draw is a pure function of (seed, env, episode, attempt), so the accepted draw — the FIRST attempt that
// passes — can be searched in any order.

// Dyn only: does draw `attempt` pass the start->end travel test (reach.py:675)?  Needs just the two positions (Philox
// blocks 1..3); 83 % of the draws fail here.
__device__ __forceinline__ bool dyn_travel_ok(const KParams& P, int n, uint32_t episode, int attempt) {
  const urgym_config& cfg = P.cfg;
  uint32_t o1[4], o2[4], o3[4];
  philox4x32_10(P.seed_lo, P.seed_hi, (uint32_t)n, episode, (uint32_t)attempt, 1u, o1);
  philox4x32_10(P.seed_lo, P.seed_hi, (uint32_t)n, episode, (uint32_t)attempt, 2u, o2);
  philox4x32_10(P.seed_lo, P.seed_hi, (uint32_t)n, episode, (uint32_t)attempt, 3u, o3);
  const double us[3] = {u01(o1[1]), u01(o1[2]), u01(o1[3])};   // u[5], u[6], u[7]
  const double ue[3] = {u01(o2[3]), u01(o3[0]), u01(o3[1])};   // u[11], u[12], u[13]
  double dd[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double a = cfg.obst_low[i] + (cfg.obst_high[i] - cfg.obst_low[i]) * us[i];
    const double b = cfg.obst_low[i] + (cfg.obst_high[i] - cfg.obst_low[i]) * ue[i];
    dd[i] = b - a;
  }
  const double d2 = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2];
  return !(sqrt(d2) < cfg.min_travel);
}

// One complete draw: goal, obstacle start (and end), every rejection test.  Returns true when the draw is REJECTED.
template <int KIND>
__device__ bool sample_attempt(const KParams& P, XRef slot, int n, uint32_t episode, int attempt, double goal[6], double st[6], double en[6]) {
  const urgym_config& cfg = P.cfg;
  const double DEG = 3.141592653589793 / 180.0;
    double u[20];
#pragma unroll
    for (int blk = 0; blk < 5; blk++) {
      uint32_t o[4];
      philox4x32_10(P.seed_lo, P.seed_hi, (uint32_t)n, episode, (uint32_t)attempt, (uint32_t)blk, o);
      u[blk * 4 + 0] = u01(o[0]); u[blk * 4 + 1] = u01(o[1]); u[blk * 4 + 2] = u01(o[2]); u[blk * 4 + 3] = u01(o[3]);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) goal[i] = cfg.goal_low[i] + (cfg.goal_high[i] - cfg.goal_low[i]) * u[i];
    if (KIND != URGYM_ENV_OBS) {  // utils.sample_euler_constrained
      goal[3] = (-90.0 + (-180.0 - -90.0) * u[3]) * DEG;
      goal[4] = 0.0 * DEG;
      goal[5] = (0.0 + (-180.0 - 0.0) * u[4]) * DEG;
    }
    if (KIND == URGYM_ENV_ORI) return false;
#pragma unroll
    for (int i = 0; i < 3; i++) st[i] = cfg.obst_low[i] + (cfg.obst_high[i] - cfg.obst_low[i]) * u[5 + i];
    {  // utils.sample_euler_obstacle
      double roll = (u[8] < 0.5) ? (-30.0 + (-150.0 - -30.0) * u[9]) : (30.0 + (150.0 - 30.0) * u[9]);
      double pitch = (roll < -90.0 || roll > 90.0) ? (-30.0 + (-150.0 - -30.0) * u[10]) : (30.0 + (150.0 - 30.0) * u[10]);
      st[3] = roll * DEG; st[4] = pitch * DEG; st[5] = 0.0 * DEG;
    }
    bool fail = false;
    // target <-> obstacle clearance (pyb_setup.py:431-437): Obs = sphere r 0.02 vs obstacle at its pose (reach.py:316-322);
    // Dyn = box half 0.025 at the goal pose vs obstacle at its END pose (reach.py:668-675)
    ShapeDesc ta;
    X3 Tt = identity_x3(), To;
    double msum;
    if (KIND == URGYM_ENV_OBS) {
      ta = point_desc();
      msum = TARGET_SPHERE_R + M_CYL;
      quat_to_rot(quat_from_rpy(st[3], st[4], st[5]), To.r);
      To.t = d3(st[0], st[1], st[2]);
    } else if (KIND == URGYM_ENV_STA) {
      // ReachSta.reset (reach.py:464-482): box target at the goal pose vs the obstacle at its sampled pose
      ta = box_desc(TARGET_BOX_H, TARGET_BOX_H, TARGET_BOX_H, M_TARGET_BOX);
      msum = M_TARGET_BOX + M_CYL;
      quat_to_rot(quat_from_rpy(st[3], st[4], st[5]), To.r);
      To.t = d3(st[0], st[1], st[2]);
      quat_to_rot(quat_from_rpy(goal[3], goal[4], goal[5]), Tt.r);
    } else {
#pragma unroll
      for (int i = 0; i < 3; i++) en[i] = cfg.obst_low[i] + (cfg.obst_high[i] - cfg.obst_low[i]) * u[11 + i];
      double roll = (u[14] < 0.5) ? (-30.0 + (-150.0 - -30.0) * u[15]) : (30.0 + (150.0 - 30.0) * u[15]);
      double pitch = (roll < -90.0 || roll > 90.0) ? (-30.0 + (-150.0 - -30.0) * u[16]) : (30.0 + (150.0 - 30.0) * u[16]);
      en[3] = roll * DEG; en[4] = pitch * DEG; en[5] = 0.0 * DEG;
      double dx = en[0] - st[0], dy = en[1] - st[1], dz = en[2] - st[2];
      fail = sqrt(dx * dx + dy * dy + dz * dz) < cfg.min_travel;
      ta = box_desc(TARGET_BOX_H, TARGET_BOX_H, TARGET_BOX_H, M_TARGET_BOX);
      msum = M_TARGET_BOX + M_CYL;
      quat_to_rot(quat_from_rpy(en[3], en[4], en[5]), To.r);
      To.t = d3(en[0], en[1], en[2]);
      quat_to_rot(quat_from_rpy(goal[3], goal[4], goal[5]), Tt.r);
    }
    if (!fail) {
      Tt.t = d3(goal[0], goal[1], goal[2]);
      int info;
      // Bullet's pair detector starts from the world +Y axis; B's frame is the obstacle's
      store(slot, rel(To, Tt));
      // Only the verdict "closer than the clearance?" is asked (reach.py:322, 675), so the search stops as soon as one of its bounds
      // decides: the support-plane distance above the limit -> clear (the returned |v| is larger still), |v| below it -> too close.
      // (the upper bound is taken a hair inside the limit, so that "<" of the reference holds for the value returned)
      const double limit = msum + cfg.target_clearance;
      double core = gjk_core_distance(P.graph, ta, slot, cyl_desc(), rotT(To, d3(0, 1, 0)), limit, info, limit * (1.0 - 1.0e-12));
      double dist = (info & GJK_PENETRATING) ? -msum : core - msum;
      fail = dist < cfg.target_clearance;
    }
    return fail;
}

// The search for the accepted draw, run by the WHOLE first wave of a RESET workgroup.  Env slot es = lane % E2 (E2 = E
// rounded up to a power of two) is led by lane es; the 64 / E2 lanes {es + k E2} of a slot test the travel rule of the
// draws base + k at once (Dyn), so the unluckiest env of a step — ~35 rejected draws among a few hundred resetting envs,
// which used to set the latency of the whole reset kernel — is through in three rounds.  The leader then evaluates the
// full draw (target <-> obstacle clearance through the GJK) and either accepts it or moves the base past it.  The result
// is the draw the sequential loop of the reference accepts; max_reset_tries bounds it the same way.
template <int KIND>
__device__ void sample_episode_wave(const KParams& P, XRef slot, int E, int lane, int n, int key, bool to_record, int& flags, int& episode_used) {
  const urgym_config& cfg = P.cfg;
  const urgym_buffers& B = P.buf;
  const int N = cfg.num_envs;
  int e2 = 1, sh = 0;
  while (e2 < E) { e2 <<= 1; sh++; }
  const int es = lane & (e2 - 1), k = lane >> sh, lpe = 64 >> sh;
  const bool leader = (lane < E) && (n >= 0);
  const int n_grp = __shfl(leader ? n : -1, es);
  // the episode id the draw is keyed with: the env's current one (live reset) or the entry's (prefetch)
  const int key_leader = leader ? (to_record ? key : B.episode_id[n]) : 0;
  const uint32_t episode = (uint32_t)__shfl(key_leader, es);
  episode_used = (int)episode;
  unsigned long long stride = 0ull;  // bit es + k E2 of a ballot belongs to slot es: shift by es, keep every E2-th bit
  for (int i = 0; i < 64; i += e2) stride |= 1ull << i;
  double goal[6] = {0, 0, 0, 0, 0, 0}, st[6] = {0, 0, 0, 0, 0, 0}, en[6] = {0, 0, 0, 0, 0, 0};
  int attempt = 0;
  bool done = !leader;
#pragma unroll 1
  for (;;) {
    if (__ballot(!done) == 0ull) break;
    if (KIND == URGYM_ENV_DYN) {
      // every slot still searching advances to its next draw that passes the travel rule; the slots meet again at the
      // (expensive) full evaluation below, so that it runs once per round for all of them
      bool cand = done;  // finished / empty slots have nothing to look for
#pragma unroll 1
      for (;;) {
        if (__ballot(!cand) == 0ull) break;
        const int base = __shfl(attempt, es);
        const bool searching = __shfl((int)!cand, es) != 0;
        const int a = base + k;
        bool pass = false;
        if (searching) {
          if (a + 1 < cfg.max_reset_tries) pass = dyn_travel_ok(P, n_grp, episode, a);
          else pass = (a + 1 == cfg.max_reset_tries);  // the last permitted draw is evaluated whatever its travel
        }
        const unsigned long long m = (__ballot(pass) >> es) & stride;
        if (!cand) {
          if (m != 0ull) { attempt = base + (__builtin_ctzll(m) >> sh); cand = true; }
          else attempt = base + lpe;
        }
      }
    }
    if (!done) {
      const bool fail = sample_attempt<KIND>(P, slot, n, episode, attempt, goal, st, en);
      if (!fail) done = true;
      else if (attempt + 1 >= cfg.max_reset_tries) { flags |= URGYM_STATUS_RESET_EXHAUSTED; done = true; }
      else attempt++;
    }
  }
  if (leader && to_record) {
    const int sl = (int)(episode & 1u);
    for (int i = 0; i < 6; i++) { REC(P, sl, REC_GOAL + i, n) = goal[i]; REC(P, sl, REC_START + i, n) = st[i]; REC(P, sl, REC_END + i, n) = en[i]; }
  } else if (leader) {
    for (int i = 0; i < 6; i++) SOA(B.goal, i, n, N) = goal[i];
    if (KIND != URGYM_ENV_ORI) {
      for (int i = 0; i < 6; i++) { SOA(B.obst_start, i, n, N) = st[i]; SOA(B.obst_end, i, n, N) = en[i]; }
    }
    for (int i = 0; i < 6; i++) SOA(B.q, i, n, N) = cfg.neutral_q[i];
    B.episode_id[n] = (int32_t)(episode + 1);
  }
}

// Joint k of env n for this launch: STEP = stored joint + float32(float32(clip(a) * pi32) * 0.1f) (UR5.py:273-279, 314);
// RESET = the neutral pose (UR5.py:327-332); REFRESH = the stored joint.  A pure function of global memory that the
// launch does not modify before its last barrier, so any lane may re-derive it instead of keeping it in LDS.
template <int MODE>
__device__ __forceinline__ double joint_of_step(const KParams& P, const float* __restrict__ actions, int n, int k) {
  if (MODE == MODE_RESET || MODE == MODE_PREFETCH) return P.cfg.neutral_q[k];
  double q = P.buf.q[(size_t)k * P.cfg.num_envs + n];
  if (MODE == MODE_STEP) {
    float a = actions[(size_t)n * 6 + k];
    a = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);   // np.clip: a NaN action stays NaN (UR5.py:275)
    float t1 = __fmul_rn(a, 3.14159274101257324f);  // float32(action * np.pi)   (UR5.py:276)
    float t2 = __fmul_rn(t1, 0.1f);                  // float32(... * 0.1)         (UR5.py:314)
    q += (double)t2;
  }
  return q;
}

// Obstacle pose of env n during a STEP launch: the stored pose advanced by this step's motion (sim.step, pyb_setup.py:52-55).
template <int KIND>
__device__ __forceinline__ void obstacle_of_step(const KParams& P, int n, double opos[3], Q4& oq) {
  const urgym_config& cfg = P.cfg;
  const urgym_buffers& B = P.buf;
  const int N = cfg.num_envs;
  for (int i = 0; i < 3; i++) opos[i] = SOA(B.obst_pos, i, n, N);
  oq = Q4{SOA(B.obst_quat, 0, n, N), SOA(B.obst_quat, 1, n, N), SOA(B.obst_quat, 2, n, N), SOA(B.obst_quat, 3, n, N)};
  if (KIND == URGYM_ENV_DYN && B.step_count[n] < cfg.dyn_motion_steps) {
    double ovel[6], dp[3];
    for (int i = 0; i < 6; i++) ovel[i] = SOA(B.obst_vel, i, n, N);
    for (int i = 0; i < 3; i++) dp[i] = SOA(B.obst_vel, 6 + i, n, N);
    integrate_obstacle(opos, oq, ovel, dp, cfg.dt);
  }
  if (KIND == URGYM_ENV_STA) {
    // core.py:307-308 + ReachSta.set_velocity (reach.py:518-541): only when obstacle_end is not all-zero; the full
    // start->end twist (time_duration = 1) while the obstacle is farther than 0.05 from its end position
    double st[6], en[6];
    bool moving = false;
    for (int i = 0; i < 6; i++) { st[i] = SOA(B.obst_start, i, n, N); en[i] = SOA(B.obst_end, i, n, N); moving = moving || en[i] != 0.0; }
    if (moving) {
      const double dx = en[0] - opos[0], dy = en[1] - opos[1], dz = en[2] - opos[2];
      double ovel[6] = {0, 0, 0, 0, 0, 0}, dp[3];
      if (sqrt(dx * dx + dy * dy + dz * dz) > 0.05) dyn_velocity(st, en, 1.0, ovel);
      step_displacement(ovel, cfg.dt, dp);
      integrate_obstacle(opos, oq, ovel, dp, cfg.dt);
    }
  }
}

// Diagnostic build only (-DURGYM_STAMPS): per-wave s_memtime stamps of the step kernel's phases (tools/phase_stamps.py).
// The stamps go to a buffer of their own; no output value depends on them.  Not compiled into the product.
#ifdef URGYM_STAMPS
#ifndef URGYM_STAMP_MODE
#define URGYM_STAMP_MODE 0  /* MODE_STEP; 1 = the auto-reset kernel */
#endif
constexpr int STAMP_BLOCKS = 8192, STAMP_SLOTS = 44;  // 0..11 phases (tools/phase_stamps.py), 12..17 + 20..23 cycles per section of the loop, 24..43 lane counters
__device__ unsigned long long g_stamps[STAMP_BLOCKS * WAVES * STAMP_SLOTS];
#define STAMP(k, v)                                                                                         \
  do {                                                                                                      \
    if (MODE == URGYM_STAMP_MODE && lane == 0 && bidx < STAMP_BLOCKS)                                              \
      g_stamps[((size_t)bidx * WAVES + wv) * STAMP_SLOTS + (k)] = (unsigned long long)(v);                  \
  } while (0)
#define STAMP_TIME(k) STAMP(k, __builtin_amdgcn_s_memtime())
#else
#define STAMP(k, v) do {} while (0)
#define STAMP_TIME(k) do {} while (0)
#endif

// LDS of one workgroup, per launch mode.  A struct (not function-local __shared__ arrays) so that the fused step launch can
// overlay the layouts of its two kinds of workgroups (STEP and PREFETCH) in one allocation.
template <int MODE>
struct EnvLds {
  // per-env slots (E envs per workgroup, <= ME) ...
  // PREFETCH workgroups run beside STEP workgroups: with at most 32 envs and no joint array they need < 53 KB as well
  static constexpr int ME = (MODE == MODE_PREFETCH) ? PREFETCH_MAX_ENVS : ((MODE == MODE_STEP) ? STEP_MAX_ENVS : MAX_ENVS);
  static constexpr bool LDS_Q = (MODE == MODE_REFRESH);  // RESET / PREFETCH: the joints are the neutral pose, a constant
  // link distances of the workgroup's envs: LDS, except STEP (global scratch [5][N], rows of consecutive envs: coalesced; the
  // cells are written by query lanes and read by the P4 lanes of the SAME workgroup after a barrier)
  static constexpr bool DIST_LDS = (MODE != MODE_STEP);
  // STEP launches re-derive the joints and the obstacle pose from global memory wherever they are needed (joint_of_step,
  // obstacle_of_step, later the set-up cache): 6.5 KB less LDS, which is what lets a third workgroup stay resident on the CU.
  // RESET / REFRESH launches hand them from the sampling lane to the query lanes through LDS.
  static constexpr bool LDS_STATE = (MODE != MODE_STEP);
  double s_dist[5][DIST_LDS ? ME : 1];
  double s_q[LDS_Q ? 6 : 1][LDS_Q ? ME : 1];                    // joint vector
  double s_obst[LDS_STATE ? 7 : 1][LDS_STATE ? ME : 1];         // obstacle position + quaternion
  uint32_t s_pairs[ME];         // culling survivors: one bit per table / track / self pair (PAIR_* below)
  int s_flags[ME];              // status bits | COLL_BIT
  int s_env[ME];                // global env id of slot e, -1 = empty slot, <= -2: non-finite joints
  // the four words of the work pool, adjacent and 16-byte aligned so that the polling of a loop trip is ONE LDS read:
  alignas(16) int s_ticket;     // next obstacle-query ticket
  int s_pending;                // number of unclaimed pair bits
  int s_p1done;                 // STEP: the per-env phase has published its pair masks
  int s_left;                   // waves that have left the work pool (the EPA service polls it)
  int s_key[(MODE != MODE_STEP) ? ME : 1];  // RESET: the env's new episode id; PREFETCH: the entry's episode
  // STEP: what a query's kind decides, as a table -- row kind | exact << 2 = {core half dims of B (3), Bullet's early-out distance, the
  // verdict distance (0: the distance itself is wanted)}: five LDS reads at the top of a trip, beside the pose, instead of sixty
  // selects and literal moves per trip that re-derived them (they may not live in registers across the search: DESIGN.md "toolchain
  // hazards" (1), and the register budget).  (The fused launch's union is sized by the PREFETCH layout: these bytes are free.)
  double s_kind[(MODE == MODE_STEP) ? 8 : 1][5];
  // ... and per-lane slots
  double s_pose[GJK_SLOT_DOUBLES][THREADS];  // GJK operand: pose of shape A in B's frame + the simplex
#ifdef URGYM_STAMPS
  unsigned long long s_clk[WAVES][PROF_WORDS];  // diagnostic build: per-wave profile of the loop (urgym_device.h: section clock + lane counters)
#endif
};

// WITH_EPA: compiled with the penetration-depth phase (the host picks the instance: a STEP launch of Dyn / Sta with the
// collision checks on can never consume a penetration depth, and its kernel stays free of that code's registers and scratch).
// bidx / nblk: index of this workgroup among those of its kind in the launch, and their number.
template <int KIND, int MODE, bool WITH_EPA>
__device__ __forceinline__ void env_body(const KParams& P, const float* __restrict__ actions, EnvLds<MODE>& L, const int bidx, const int nblk) {
  constexpr int ME = EnvLds<MODE>::ME;
  constexpr bool LDS_Q = EnvLds<MODE>::LDS_Q, DIST_LDS = EnvLds<MODE>::DIST_LDS, LDS_STATE = EnvLds<MODE>::LDS_STATE;
  auto& s_dist = L.s_dist;
  auto& s_q = L.s_q;
  auto& s_obst = L.s_obst;
  auto& s_pairs = L.s_pairs;
  auto& s_flags = L.s_flags;
  auto& s_env = L.s_env;
  int& s_ticket = L.s_ticket;
  int& s_pending = L.s_pending;
  int& s_p1done = L.s_p1done;
  int& s_left = L.s_left;
  auto& s_key = L.s_key;
  auto& s_pose = L.s_pose;

  const urgym_config& cfg = P.cfg;
  const urgym_buffers& B = P.buf;
  const int N = cfg.num_envs;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // observation / goal widths (core.py:241-247): compile-time per env kind, so that the row loops unroll and no per-lane
  // array is indexed dynamically (that would live in scratch)
  constexpr int OD = (KIND == URGYM_ENV_ORI) ? 18 : ((KIND == URGYM_ENV_OBS) ? 26 : ((KIND == URGYM_ENV_STA) ? 29 : 35));
  constexpr int GD = (KIND == URGYM_ENV_OBS) ? 3 : 6;
  // envs of this workgroup (1 .. ME) and the index of its first env / list entry.  STEP launches may be two-tiered.
  // XCD-aware workgroup -> env-range mapping (STEP, uniform geometry): the hardware deals workgroups round-robin over the 8 XCDs
  // (workgroup b runs on XCD b % 8), each with its own L2.  Workgroup b therefore serves env range vb = start(b % 8) + b / 8, where
  // XCD x owns the contiguous block of ranges [start(x), start(x + 1)): neighbouring ranges -- which share the 128-byte lines at
  // the ends of their runs in the float64 state arrays -- then meet in ONE L2 instead of being fetched by two.
  int vb = bidx;
  if (MODE == MODE_STEP && P.envs_tail == 0) {
    const int nb = nblk, x = vb & 7, per = nb >> 3, rem = nb & 7;
    vb = x * per + (x < rem ? x : rem) + (vb >> 3);
  }
  const bool tail_block = (MODE == MODE_STEP) && P.envs_tail > 0 && vb >= P.big_blocks;
  const int E = tail_block ? P.envs_tail : P.envs;
  const int first = tail_block ? P.big_blocks * P.envs + (vb - P.big_blocks) * P.envs_tail : vb * P.envs;
  const int G = (E + GROUP - 1) / GROUP;     // waves that run the per-env phases
  constexpr bool HAS_OBST = (KIND != URGYM_ENV_ORI);
  constexpr int COLL_BIT = 1 << 30;
  constexpr int EE_BIT = 1 << 29;  // STEP: the per-env phase has already stored the end-effector pose of this step (P4 reads it back)
  // bits 8..22 of s_flags: exact queries that ended with overlapping cores and whose distance IS consumed -> penetration depth
  // by EPA after the pool has drained; bit = 5 * body + (link - 2), body 0 obstacle, 1 table, 2 track (WORKBENCH scope)
  constexpr int EPA_SHIFT = 8, EPA_MASK = 0x7FFF << EPA_SHIFT;
  // Dyn / Sta return -500 on a collision before the distances are used (reach.py:766-767), and overlapping cores ARE a
  // collision: only Obs (reach.py:357-372), the collision-free variant and reset / refresh consume a penetration depth
  const bool need_epa = WITH_EPA && ((KIND == URGYM_ENV_OBS) || !cfg.check_collision || (MODE != MODE_STEP));
  XRef pose_slot{(URGYM_LDS double*)&s_pose[0][0] + tid, THREADS};
  float* const s_out = reinterpret_cast<float*>(&s_pose[0][0]);  // observation rows are staged here once the GJK slots are free
  static_assert(sizeof(float) * STEP_MAX_ENVS * 47 <= sizeof(double) * GJK_SLOT_DOUBLES * THREADS, "staging must fit");
  auto dist_cell = [&](int i, int e) -> double* {
    return DIST_LDS ? (double*)&s_dist[i][DIST_LDS ? e : 0] : &P.ld_scratch[(size_t)i * N + first + e];
  };

  if (bidx == 0 && tid == 0) {  // (before any early exit)
    if (P.rzero) *P.rzero = 0;
    if (P.rzero2) *P.rzero2 = 0;
  }
  int list_count = 0;
  if (MODE != MODE_STEP) {
    list_count = (MODE == MODE_PREFETCH) ? min(*P.rcount, P.rcap) : B.done_count[P.pp];
    if (first >= list_count) return;  // uniform for the whole workgroup
  }
  if (tid == 0) { s_ticket = THREADS; s_pending = 0; s_p1done = 0; s_left = 0; }
  if (MODE == MODE_STEP && tid < 8) {
    const int k = tid & 3, ex = tid >> 2;
    const bool tbl = (k == Q_TABLE);
    const double msum = M_HULL + ((k == 3) ? M_CYL : ((k == Q_SELF) ? M_HULL : (tbl ? M_TABLE : M_TRACK)));
    const bool wants = (k == 3) || ex;
    const double verdict = wants ? 0.0 : msum + cfg.collision_margin;
    L.s_kind[tid][0] = (k == 3) ? (CYL_R - M_CYL) : (tbl ? (TABLE_HX - M_TABLE) : (TRACK_HX - M_TRACK));
    L.s_kind[tid][1] = (k == 3) ? (CYL_R - M_CYL) : (tbl ? (TABLE_HY - M_TABLE) : (TRACK_HY - M_TRACK));
    L.s_kind[tid][2] = (k == 3) ? (0.5 * CYL_H - M_CYL) : (tbl ? (TABLE_HZ - M_TABLE) : (TRACK_HZ - M_TRACK));
    L.s_kind[tid][3] = wants ? msum + 0.02 + 5.0 : verdict;
    L.s_kind[tid][4] = verdict;
  }
  // URGYM_LINK_DIST_WORKBENCH: link_dist[i] = min over obstacle, table, track -- three exact queries per link race for the
  // cell, which therefore holds the order-preserving int64 image of the distance (sortable()); +inf is its own image
  const bool workbench = (KIND != URGYM_ENV_ORI) && cfg.link_dist_scope == URGYM_LINK_DIST_WORKBENCH;
  if (workbench && tid < E && (DIST_LDS || first + tid < N))  // (the global scratch has exactly N cells per row)
    for (int i = 0; i < 5; i++) *dist_cell(i, tid) = __longlong_as_double(0x7FF0000000000000LL);
  // STEP: the per-env phase P1 (joint check + culling) runs on the LAST wave while the others already start their obstacle
  // queries — nothing a query needs comes from P1 (joints and obstacle are re-derived from global memory), only the pair
  // masks do, and those are drawn late.  So the slots are initialised here, before the first barrier, and the barrier after
  // P1 is dropped for STEP.  The last wave holds the fewest / shortest queries (ticket order), which hides its late start.
  // (with E > 64 the last TWO waves: wave WAVES - G + g serves env slots 64 g .. 64 g + 63, the same lanes as in P4)
  if (MODE == MODE_STEP && tid < E) {
    s_env[tid] = (first + tid < N) ? first + tid : -1;
    s_flags[tid] = 0;
    s_pairs[tid] = 0;
  }
  STAMP_TIME(0);
  STAMP(8, __builtin_amdgcn_s_memrealtime());
#ifdef URGYM_STAMPS
  { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); STAMP(10, hw); STAMP(11, xcc); }
#endif
  __syncthreads();
  // ---- P1 (waves 0..G-1, one lane per env slot): which env, joint update, obstacle motion, and the conservative
  //      bounding-capsule culling of the table / track / self pairs of check_collision (pyb_setup.py:407-427) -> LDS
  const int p1_slot = (MODE == MODE_STEP) ? (wv - (WAVES - G)) * GROUP + lane : lane;  // env slot this lane serves in P1
  const bool p1_lane = ((MODE == MODE_STEP) ? (wv >= WAVES - G) : (wv == 0)) && (p1_slot < E);
  int n_slot = -1, flags_slot = 0;  // env of slot `lane`
  int key_slot = 0;                 // PREFETCH: the episode id the record is for
  if (p1_lane) {
    const int idx = first + p1_slot;
    if (MODE == MODE_STEP) n_slot = idx < N ? idx : -1;
    else if (MODE == MODE_PREFETCH) {
      if (idx < list_count) { const int2 ent = P.rlist[idx]; n_slot = ent.x; key_slot = ent.y; }
    } else n_slot = idx < list_count ? B.done_list[idx] : -1;
  }
  // RESET / PREFETCH: the whole first wave searches the accepted draws of its (<= 64) env slots.  RESET writes goal /
  // obstacle / q / episode_id of the live state, PREFETCH the goal / obstacle fields of the record slot.
  if ((MODE == MODE_RESET || MODE == MODE_PREFETCH) && wv == 0) {
    int episode_used = 0;
    sample_episode_wave<KIND>(P, pose_slot, E, lane, n_slot, key_slot, MODE == MODE_PREFETCH, flags_slot, episode_used);
    if (p1_lane) s_key[p1_slot] = (MODE == MODE_PREFETCH) ? key_slot : episode_used + 1;
  }
  if (p1_lane) {
    const int e = p1_slot;
    const int n = n_slot;
    int flags = flags_slot;
    double q[6] = {0, 0, 0, 0, 0, 0};
    double opos[3] = {0, 0, 0};
    Q4 oq{0, 0, 0, 1};
    bool finite = true;
    if (n >= 0) {
      for (int i = 0; i < 6; i++) q[i] = joint_of_step<MODE>(P, actions, n, i);
      for (int i = 0; i < 6; i++) finite = finite && (fabs(q[i]) < 1.0e6);  // false for NaN/inf: no distance queries then
      if (MODE == MODE_STEP) {
        // resetJointState does not clamp (pyb_setup.py:338), so neither does this; but past the URDF limit (ur5e.urdf:237-277:
        // elbow +-pi, the others +-2 pi) Bullet's limit constraint would act during stepSimulation -- flagged, not altered
        bool over = false;
        for (int i = 0; i < 6; i++) over = over || (fabs(q[i]) > (i == 2 ? 3.141592653589793 : 6.283185307179586));
        if (over) atomicOr(&s_flags[p1_slot], URGYM_STATUS_JOINT_LIMIT);
      }
      if (HAS_OBST && LDS_STATE) {
        // reset / refresh: the obstacle goes to its start pose (reach.py:319, 678, 709-710); PREFETCH: the record's
        double sp[6];
        for (int i = 0; i < 6; i++) sp[i] = (MODE == MODE_PREFETCH) ? REC(P, key_slot & 1, REC_START + i, n) : SOA(B.obst_start, i, n, N);
        for (int i = 0; i < 3; i++) opos[i] = sp[i];
        oq = quat_from_rpy(sp[3], sp[4], sp[5]);
      }
    }
    const bool live = (n >= 0 && finite);
    // (the set-up cache serves the draws of closest-distance queries: a launch that runs none -- UR5OriReach-v1 with the collision checks
    //  off -- does not fill it; its six sin / cos evaluations were a third of that launch's span)
    const bool use_cache = (MODE == MODE_STEP) && P.sc_scratch != nullptr && (HAS_OBST || cfg.check_collision);
    if (use_cache && live) {  // the set-up cache of this env (KParams::sc_scratch)
#pragma unroll 1
      for (int k = 0; k < 6; k++) {
        double sn, cs;
        sincos(joint_of_step<MODE>(P, actions, n, k), &sn, &cs);
        SOA(P.sc_scratch, 2 * k, n, N) = sn;
        SOA(P.sc_scratch, 2 * k + 1, n, N) = cs;
      }
      if (HAS_OBST) {
        obstacle_of_step<KIND>(P, n, opos, oq);
        for (int i = 0; i < 3; i++) SOA(P.sc_scratch, 12 + i, n, N) = opos[i];
        SOA(P.sc_scratch, 15, n, N) = oq.x; SOA(P.sc_scratch, 16, n, N) = oq.y; SOA(P.sc_scratch, 17, n, N) = oq.z; SOA(P.sc_scratch, 18, n, N) = oq.w;
      }
    }
    s_env[e] = live ? n : (n >= 0 ? -2 - n : -1);  // -1 empty; <= -2: env (-2 - v) with non-finite joints
    if (MODE != MODE_STEP) s_flags[e] = flags;      // (STEP: zeroed before the first barrier; queries may already be OR-ing)
    if (LDS_STATE) {
      if (LDS_Q) for (int i = 0; i < 6; i++) s_q[i][e] = q[i];
      s_obst[0][e] = opos[0]; s_obst[1][e] = opos[1]; s_obst[2][e] = opos[2];
      s_obst[3][e] = oq.x; s_obst[4][e] = oq.y; s_obst[5][e] = oq.z; s_obst[6][e] = oq.w;
    }
    if (HAS_OBST && !live && (DIST_LDS || n >= 0))  // (empty slots have no cell in the global scratch; P4 skips them)
      for (int i = 0; i < 5; i++) *dist_cell(i, e) = (n >= 0) ? __builtin_nan("") : 1e30;
    // culling: one FK pass over the six links, world bounding capsules, segment-box / segment-segment lower bounds
    uint32_t pairs = 0;
    if (live && cfg.check_collision && MODE != MODE_RESET && MODE != MODE_PREFETCH) {
      // The capsules bound the hull VERTICES; Bullet's hull is those vertices inflated by its margin (the boxes are shrunk cores plus
      // their margin, i.e. inside the full boxes used here), so a pair can be closer than its capsules by one hull margin (two for
      // a self pair).  Round 2 found the bound without that term: a table contact at 0.00988 m was culled (margin 0.01).
      const double lim = cfg.collision_margin + M_HULL + 1e-6;
      X3 T = identity_x3();
      D3 a0[3], a1[3];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        double sn, cs;
        if (use_cache) { sn = SOA(P.sc_scratch, 2 * k, n, N); cs = SOA(P.sc_scratch, 2 * k + 1, n, N); }  // (this lane stored them above)
        else sincos(q[k], &sn, &cs);
        fk_joint(T, k, sn, cs);
        if (use_cache && P.sc_frames) {  // the frame of link k + 1: later draws read it instead of multiplying the chain up again
#pragma unroll
          for (int j = 0; j < 9; j++) SOA(P.sc_scratch, SC_FRAMES + 12 * k + j, n, N) = T.r[j];
          SOA(P.sc_scratch, SC_FRAMES + 12 * k + 9, n, N) = T.t.x;
          SOA(P.sc_scratch, SC_FRAMES + 12 * k + 10, n, N) = T.t.y;
          SOA(P.sc_scratch, SC_FRAMES + 12 * k + 11, n, N) = T.t.z;
        }
        const int link = k + 1;
        const double* c = c_tab.capsule[k];
        const D3 b0 = apply(T, d3(c[0], c[1], c[2])), b1 = apply(T, d3(c[3], c[4], c[5]));
        const double rb = c[6];
        if (k < 3) { a0[k] = b0; a1[k] = b1; }
        if (link >= 2) {
          if (seg_box_lower_bound(b0, b1, TABLE_CX, TABLE_CY, TABLE_CZ, TABLE_HX, TABLE_HY, TABLE_HZ) - rb <= lim) pairs |= 1u << (PAIR_TABLE + link - 2);
          if (seg_box_lower_bound(b0, b1, TRACK_CX, TRACK_CY, TRACK_CZ, TRACK_HX, TRACK_HY, TRACK_HZ) - rb <= lim) pairs |= 1u << (PAIR_TRACK + link - 2);
        }
        // self pairs (pyb_setup.py:417-427): (1,3)(1,4)(1,5)(1,6)(2,4)(2,5)(2,6)(3,5)(3,6)
#pragma unroll
        for (int A = 1; A <= 3; A++) {
          if (A <= link - 2) {
            if (segseg_dist(a0[A - 1], a1[A - 1], b0, b1) - c_tab.capsule[A - 1][6] - rb <= lim + M_HULL) pairs |= 1u << self_pair_bit(A, link);
          }
        }
      }
      if (MODE == MODE_STEP) {
        // T is now the end-effector frame (link 6 == ee_link 7): read it out here, in the shadow of the other waves' queries,
        // instead of repeating the FK in P4 on the workgroup's critical path.  This very lane is also the env's P4 lane; the
        // six float32 values are parked in the env's observation row (their final place) and read back there.
        double er, ep, ey;
        rpy_from_quat(rot_to_quat(T.r), er, ep, ey);
        float* orow = B.observation + (size_t)n * OD;
        orow[0] = (float)T.t.x; orow[1] = (float)T.t.y; orow[2] = (float)T.t.z;
        orow[3] = (float)er; orow[4] = (float)ep; orow[5] = (float)ey;
        atomicOr(&s_flags[e], EE_BIT);
      }
    }
    s_pairs[e] = pairs;
    if (pairs) atomicAdd(&s_pending, __popc(pairs));
  }
  STAMP_TIME(1);
  if (MODE == MODE_STEP) {
    if (wv >= WAVES - G && lane == 0) __hip_atomic_fetch_add(&s_p1done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    __syncthreads();
  }

  // ---- P2 + P3: all closest-distance work of the workgroup goes through ONE inlined, resumable GJK body, fed from a
  //      work pool so that no lane waits for the slowest query of its wave:
  //   obstacle tickets t in [0, 5E): exact distance hull(link 2 + t / E) <-> obstacle cylinder of env slot t % E
  //             (pyb_setup.py:439-456; also the obstacle rule of check_collision).  The first 256 tickets are the lanes'
  //             own (with E = 64: one hull per wave), the rest (E > 51) is drawn from s_ticket as lanes finish.
  //   pair bits: the culling survivors of P1, boolean "closer than the margin?" queries, claimed bit by bit once the
  //             tickets are gone.
  //   A lane advances its query by one GJK iteration per loop trip; a finished lane draws the next item.
  // (A launch that can have no query at all -- no obstacle, collision checks off: BASELINE configs[1] "FK + reward only" -- skips the
  //  pool, its polling trip and its barrier: P4 runs on the very lanes that ran P1.)
  if (!(MODE == MODE_STEP && !HAS_OBST && !cfg.check_collision)) {
    // The operands of a lane's query are carried as (kind, la, lb) only; the shape descriptors are re-derived from them at
    // each use (a handful of selects) instead of living in 16 VGPRs across the loop, which is what used to push the sincos
    // constants of the set-up into scratch.
    D3 v0 = d3(0, 1, 0);
    int kind = 3, e = 0, lb = 2, la = 1;
    auto shape_a = [&]() -> ShapeDesc { return hull_desc(kind == Q_SELF ? la : lb); };
    auto shape_b = [&]() -> ShapeDesc {
      ShapeDesc s;
      s.type = (kind == 3) ? SH_CYLZ : ((kind == Q_SELF) ? SH_HULL : SH_BOX);
      s.hull = lb - 1;
      const bool tbl = (kind == Q_TABLE);
      s.hx = (kind == 3) ? (CYL_R - M_CYL) : (tbl ? (TABLE_HX - M_TABLE) : (TRACK_HX - M_TRACK));
      s.hy = (kind == 3) ? (CYL_R - M_CYL) : (tbl ? (TABLE_HY - M_TABLE) : (TRACK_HY - M_TRACK));
      s.hz = (kind == 3) ? (0.5 * CYL_H - M_CYL) : (tbl ? (TABLE_HZ - M_TABLE) : (TRACK_HZ - M_TRACK));
      return s;
    };
    bool exact = false;  // table / track item that wants the distance itself (WORKBENCH link_dist), not just "closer than the margin?"
    // URGYM_GJK_START_GUIDED (include/urgym.h): first separating axis = unit vector from the other shape's centre to
    // the mid point of the link's bounding capsule; same arithmetic as the oracle's guided_axis()
    const bool guided = cfg.gjk_start == URGYM_GJK_START_GUIDED;
    auto capsule_mid = [&](int link, const X3& TX) -> D3 {
      const double* c = c_tab.capsule[link - 1];
      return apply(TX, d3(0.5 * (c[0] + c[3]), 0.5 * (c[1] + c[4]), 0.5 * (c[2] + c[5])));
    };
    auto guided_axis = [&](D3 mid, D3 centre) -> D3 {
      const D3 d = mid - centre;
      const double n2 = dot(d, d);
      return n2 > 1e-12 ? d * (1.0 / sqrt(n2)) : d3(0, 1, 0);
    };
    // builds the operands of one work item (e | kind << 8 | lb << 10 | la << 13); false when there is nothing to run
    // p1_ok: the P1 lanes of this workgroup have published (s_p1done acquired) -- their set-up cache may be read
    bool p1_ok = false;
    auto setup = [&](uint32_t item) -> bool {
      e = item & 255;
      kind = (item >> 8) & 3;
      exact = ((item >> 16) & 1) != 0;
      lb = (item >> 10) & 7;
      la = (item >> 13) & 7;
      // An item is decoded from a ticket number or a pair bit.  Whatever produced it, nothing below may index with fields that are
      // out of range: e picks the env slot (s_env, the per-env rows of the global scratch arrays), lb / la pick hull tables.  (Round 2
      // lost a GPU to an experiment whose ticket-order list fed this decoder -- profiles/r3/EXPERIMENTS.md; the order is gone, the
      // bound stays.)
      if (e >= E || lb < 1 || lb > 6 || (kind == Q_SELF && (la < 1 || la > 6))) return false;
      const int n = s_env[e];
      if (n < 0) return false;
      const bool cached = (MODE == MODE_STEP) && p1_ok && P.sc_scratch != nullptr;  // (then s_env already carries P1's verdict on the joints)
      if (MODE == MODE_STEP && !cached) {  // P1 may not have judged this env yet: non-finite joints -> no query (same rule as P1)
        bool finite = true;
        for (int k = 0; k < 6; k++) finite = finite && (fabs(joint_of_step<MODE>(P, actions, n, k)) < 1.0e6);
        if (!finite) return false;
      }
      X3 T = identity_x3(), TA = identity_x3();
      if (cached && P.sc_frames && cfg.check_collision) {  // P1's culling pass left the link frames in the cache (same fk_joint chain, same bits)
        auto frame = [&](int link, X3& F) {
#pragma unroll
          for (int j = 0; j < 9; j++) F.r[j] = SOA(P.sc_scratch, SC_FRAMES + 12 * (link - 1) + j, n, N);
          F.t = d3(SOA(P.sc_scratch, SC_FRAMES + 12 * (link - 1) + 9, n, N), SOA(P.sc_scratch, SC_FRAMES + 12 * (link - 1) + 10, n, N),
                   SOA(P.sc_scratch, SC_FRAMES + 12 * (link - 1) + 11, n, N));
        };
        frame(lb, T);
        if (kind == Q_SELF) frame(la, TA);
      } else {
#pragma unroll 1
        for (int k = 0; k < lb; k++) {
          double sn, cs;
          if (cached) { sn = SOA(P.sc_scratch, 2 * k, n, N); cs = SOA(P.sc_scratch, 2 * k + 1, n, N); }
          else sincos(LDS_Q ? s_q[k][e] : joint_of_step<MODE>(P, actions, n, k), &sn, &cs);
          fk_joint(T, k, sn, cs);
          if (k + 1 == la) TA = T;
        }
      }
      if (kind == 3) {
        X3 To;
        if (LDS_STATE) {
          quat_to_rot(Q4{s_obst[3][e], s_obst[4][e], s_obst[5][e], s_obst[6][e]}, To.r);
          To.t = d3(s_obst[0][e], s_obst[1][e], s_obst[2][e]);
        } else if (cached) {
          quat_to_rot(Q4{SOA(P.sc_scratch, 15, n, N), SOA(P.sc_scratch, 16, n, N), SOA(P.sc_scratch, 17, n, N), SOA(P.sc_scratch, 18, n, N)}, To.r);
          To.t = d3(SOA(P.sc_scratch, 12, n, N), SOA(P.sc_scratch, 13, n, N), SOA(P.sc_scratch, 14, n, N));
        } else {
          double op[3];
          Q4 oqs;
          obstacle_of_step<KIND>(P, n, op, oqs);
          quat_to_rot(oqs, To.r);
          To.t = d3(op[0], op[1], op[2]);
        }
        store(pose_slot, rel(To, T));
        v0 = rotT(To, guided ? guided_axis(capsule_mid(lb, T), To.t) : d3(0, 1, 0));  // Bullet: the world +Y axis
      } else if (kind == Q_SELF) {
        store(pose_slot, rel(T, TA));
        v0 = rotT(T, guided ? guided_axis(capsule_mid(la, TA), capsule_mid(lb, T)) : d3(0, 1, 0));
      } else {
        const bool tbl = (kind == Q_TABLE);
        const D3 centre = d3(tbl ? TABLE_CX : TRACK_CX, tbl ? TABLE_CY : TRACK_CY, tbl ? TABLE_CZ : TRACK_CZ);
        v0 = guided ? guided_axis(capsule_mid(lb, T), centre) : d3(0, 1, 0);
        T.t = T.t - centre;
        store(pose_slot, T);
      }
      return true;
    };
    // claims one set bit of s_pairs (the caller holds a claim on s_pending, so one exists) and decodes it into an item
    auto claim_pair = [&]() -> uint32_t {
      int pe = tid % E;
#pragma unroll 1
      for (int trip = 0; trip < 2 * ME; trip++, pe = (pe + 1 == E ? 0 : pe + 1)) {
        uint32_t m = s_pairs[pe];
        while (m) {
          const int b = __ffs((int)m) - 1;
          const uint32_t old = atomicAnd(&s_pairs[pe], ~(1u << b));
          if (old & (1u << b)) {
            if (b < PAIR_TRACK) return (uint32_t)pe | (Q_TABLE << 8) | ((uint32_t)(b - PAIR_TABLE + 2) << 10);
            if (b < PAIR_SELF) return (uint32_t)pe | (Q_TRACK << 8) | ((uint32_t)(b - PAIR_TRACK + 2) << 10);
            const int i = b - PAIR_SELF;  // (1,3)(1,4)(1,5)(1,6)(2,4)(2,5)(2,6)(3,5)(3,6)
            const int la = i < 4 ? 1 : (i < 7 ? 2 : 3);
            const int l2 = i < 4 ? 3 + i : (i < 7 ? i : i - 2);
            return (uint32_t)pe | (Q_SELF << 8) | ((uint32_t)l2 << 10) | ((uint32_t)la << 13);
          }
          m = old & ~(1u << b);
        }
      }
      return NO_ITEM;  // unreachable while the claim count is right; bounded so that a wave can never spin here
    };
    // Bullet margins of the pair and its early-out distance (margins + 0.02 + query threshold): get_link_distances
    // queries with distance=5.0 (pyb_setup.py:452), check_collision with 0.01 (pyb_setup.py:402-422).  Recomputed from
    // `kind` where needed: nothing constant-like stays live across the search (see DESIGN.md "toolchain hazard").
    auto margin_sum = [&]() -> double {
      return M_HULL + ((kind == 3) ? M_CYL : ((kind == Q_SELF) ? M_HULL : ((kind == Q_TABLE) ? M_TABLE : M_TRACK)));
    };
    const int n_tickets = HAS_OBST ? (workbench ? 15 : 5) * E : 0;
    auto ticket_item = [&](int t) -> uint32_t {
      const int body = t / (5 * E), r = t - body * 5 * E;  // 0 obstacle, then (WORKBENCH) 1 table, 2 track
      const int tl = r / E, te = r - tl * E, link = 2 + tl;
      if (body == 0) return (uint32_t)te | (3u << 8) | ((uint32_t)link << 10);
      return (uint32_t)te | ((uint32_t)(body == 1 ? Q_TABLE : Q_TRACK) << 8) | ((uint32_t)link << 10) | (1u << 16);
    };

    GjkRun run;
    bool busy = false;
    STAMP_TIME(2);
    if (MODE == MODE_STEP) p1_ok = __hip_atomic_load(&s_p1done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= G;
    if (tid < n_tickets) {
      busy = setup(ticket_item(tid));
      if (busy) gjk_begin(run, v0);
    }
    STAMP_TIME(3);
    int trips = 0, draws = 0;
#ifdef URGYM_STAMPS
    // cycles of this wave per section of a loop trip (a section ends at its mark): 1 support of A, 2 support of B + exits, 3 simplex +
    // reduction + convergence tests, 4 result handling, 5 polling + draw + set-up (a mark that no lane of a trip reaches adds its
    // time to the next section)
    URGYM_LDS unsigned long long* clk = (URGYM_LDS unsigned long long*)&L.s_clk[wv][0];
    if (lane < PROF_WORDS) clk[lane] = 0;
    if (busy) run.clk = clk;
    clk[0] = __builtin_amdgcn_s_memtime();
#define SECTION(i) trip_mark(clk, i)
#else
#define SECTION(i) do {} while (0)
#endif
#ifdef URGYM_STAMPS
    unsigned long long n_boxq = 0, n_selfq = 0, trips_self = 0;  // census of the pair queries this wave ran / trips that carried a hull <-> hull query
#endif
    for (;;) {
      trips++;
#ifdef URGYM_STAMPS
      trips_self += (__ballot(busy && kind == Q_SELF) != 0ull) ? 1 : 0;
      {  // counter 0: one execution per trip, lanes = the busy ones
        const unsigned long long bm = __ballot(busy);
        if (lane == 0) { URGYM_LDS unsigned int* w = (URGYM_LDS unsigned int*)(clk + 1 + PROF_SECTIONS); w[0] += 1u; w[1] += (unsigned int)__popcll(bm); }
      }
#endif
      if (busy) {
        // exact queries (link distances): Bullet's early-out distance of getClosestPoints(distance = 5.0).  Boolean queries ("closer
        // than the contact margin?", check_collision): both bounds of the search are compared with the margin itself -- the search
        // stops as soon as either decides, with the verdict Bullet reaches after converging (pyb_setup.py:402-422)
        if (MODE == MODE_STEP) {
          const URGYM_LDS double* row = (const URGYM_LDS double*)(URGYM_LDS void*)&L.s_kind[kind | ((int)exact << 2)][0];
          ShapeDesc sb;
          sb.type = (kind == 3) ? SH_CYLZ : ((kind == Q_SELF) ? SH_HULL : SH_BOX);
          sb.hull = lb - 1;
          sb.hx = row[0]; sb.hy = row[1]; sb.hz = row[2];
          gjk_iterate(run, P.graph, shape_a(), pose_slot, sb, row[3], row[4]);
        } else {
          const bool wants_distance = (kind == 3 || exact);
          const double verdict_d = wants_distance ? 0.0 : margin_sum() + cfg.collision_margin;
          gjk_iterate(run, P.graph, shape_a(), pose_slot, shape_b(), wants_distance ? margin_sum() + 0.02 + 5.0 : verdict_d, verdict_d);
        }
        SECTION(3);
        if (run.done) {
#ifdef URGYM_STAMPS
          lane_mark(clk, 15);
#endif
          const double msum = margin_sum();
          if (kind == 3 || exact) {
            double dist = run.core - msum;
            if (run.info & GJK_PENETRATING) {
              // provisional value (exact only when the cores just touch); the EPA phase below replaces it where it is consumed
              dist = -msum;
              const int body = (kind == 3) ? 0 : (kind == Q_TABLE ? 1 : 2);
              atomicOr(&s_flags[e], URGYM_STATUS_PENETRATION | (need_epa ? (1 << (EPA_SHIFT + 5 * body + (lb - 2))) : 0));
            }
            if (run.info & GJK_ITERCAP) atomicOr(&s_flags[e], URGYM_STATUS_GJK_ITER);
            if (workbench) atomicMin(reinterpret_cast<long long*>(dist_cell(lb - 2, e)), sortable(dist));
            else *dist_cell(lb - 2, e) = dist;
          } else {
            const bool hit = (run.info & (GJK_PENETRATING | GJK_CLOSE)) || (!(run.info & GJK_SEPARATED) && (run.core - msum) <= cfg.collision_margin);
            if (hit) atomicOr(&s_flags[e], COLL_BIT);
          }
          busy = false;
        }
      }
      SECTION(4);
      // (atomic loads: other waves change both words while this one polls them; a plain read could legally be hoisted)
      // one 16-byte LDS read of the pool's words (three dependent round trips when they were polled one by one), then the acquire:
      // what the P1 lanes wrote before they counted themselves in s_p1done (pair masks, the set-up cache) is visible to what follows
      typedef int pool_words __attribute__((ext_vector_type(4)));
      const pool_words pool = *(volatile URGYM_LDS pool_words*)(URGYM_LDS void*)&s_ticket;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const bool more_tickets = pool.x < n_tickets;
      const bool p1_published = (MODE != MODE_STEP) || pool.z >= G;
      const bool more_pairs = pool.y > 0;
      if (MODE == MODE_STEP) p1_ok = p1_published;
      // drawing an item costs the whole wave a set-up (FK + operands), so idle lanes
      // draw together: once REFILL_MIN of them are waiting, or when none is busy any more
      const int idle_lanes = __popcll(__ballot(!busy));
      if (!busy && (idle_lanes >= REFILL_MIN || idle_lanes == 64)) {
        uint32_t item = NO_ITEM;
        if (more_tickets) {
          const int t = atomicAdd(&s_ticket, 1);
          if (t < n_tickets) item = ticket_item(t);
        } else if (more_pairs) {
          if (atomicSub(&s_pending, 1) > 0) item = claim_pair();
        }
        if (item != NO_ITEM) {
          draws++;  // (diagnostic: the draws of the lane that reports the stamps)
#ifdef URGYM_STAMPS
          lane_mark(clk, 14);
#endif
          busy = setup(item);
          if (busy) gjk_begin(run, v0);
#ifdef URGYM_STAMPS
          if (busy) run.clk = clk;
          n_boxq += __popcll(__ballot(busy && (kind == Q_TABLE || kind == Q_TRACK)));
          n_selfq += __popcll(__ballot(busy && kind == Q_SELF));
#endif
        }
      }
      SECTION(5);
      // nothing left for this wave to draw (STEP: and the pair masks have been published)
      if (__ballot(busy) == 0ull && !more_tickets && !more_pairs && p1_published) break;
    }
#ifdef URGYM_STAMPS
    for (int i = 0; i < 6; i++) STAMP(12 + i, clk[1 + i]);
    for (int i = 6; i < PROF_SECTIONS; i++) STAMP(20 + i - 6, clk[1 + i]);
    for (int i = 0; i < PROF_COUNTERS; i++) STAMP(24 + i, clk[1 + PROF_SECTIONS + i]);
    STAMP(18, n_boxq | (n_selfq << 32));
    STAMP(19, trips_self);
#endif
    STAMP_TIME(4);
    STAMP(5, (unsigned long long)trips | ((unsigned long long)draws << 32));
    (void)trips; (void)draws;  // (diagnostic counters of the stamps build)
    // ---- EPA: penetration depth of the marked queries, one wave per query (urgym_device.h epa_wave).  A wave that has left
    //      the pool turns into a service wave: it keeps looking for marks and serves them while the other waves still iterate
    //      (overlapping cores are found within a few GJK iterations, an EPA takes 70-250 us: starting it at once instead of
    //      after the pool has drained hides most of it), until every wave has left the pool and no mark is unclaimed.  A mark is
    //      claimed by clearing its bit, so each is served exactly once.
    auto epa_pass = [&]() -> bool {
      const EpaWs ws{(URGYM_LDS double*)&s_pose[0][0] + GROUP * wv, THREADS};
      bool armed = false;
#pragma unroll 1
      for (int g = 0; g < G; g++) {
        const int sl = g * GROUP + lane;
        const int marks = (sl < E) ? ((__hip_atomic_load(&s_flags[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & EPA_MASK) >> EPA_SHIFT) : 0;
        unsigned long long envs_marked = __ballot(marks != 0);
#pragma unroll 1
        while (envs_marked) {
          const int el = __builtin_ctzll(envs_marked);
          envs_marked &= envs_marked - 1ull;
          const int ee = g * GROUP + el;
          int me = __shfl(marks, el);
#pragma unroll 1
          while (me) {
            const int b = __builtin_ctz((unsigned)me);
            me &= me - 1;
            int mine = 0;
            if (lane == 0) mine = (atomicAnd(&s_flags[ee], ~(1 << (EPA_SHIFT + b))) >> (EPA_SHIFT + b)) & 1;
            if (!__shfl(mine, 0)) continue;  // another wave took it
            if (!armed) {  // every lane of the wave stores the (same) operands into the wave's slot
              pose_slot.p = ws.base;
              if (MODE == MODE_STEP) p1_ok = true;  // (a mark exists only once P1 has published: its query was drawn after it ... or ran without the cache)
              armed = true;
            }
            const int body = b / 5, link = 2 + b - 5 * body;
            const uint32_t item = (uint32_t)ee | ((uint32_t)(body == 0 ? 3 : (body == 1 ? Q_TABLE : Q_TRACK)) << 8) | ((uint32_t)link << 10) |
                                  ((body ? 1u : 0u) << 16);
            if (!setup(item)) continue;
            epa_wave_sync();
            bool capped;
            const double depth = epa_wave(P.graph, shape_a(), shape_b(), ws, lane, capped);
            if (lane == 0) {
              const double dist = -(depth + margin_sum());
              if (workbench) atomicMin(reinterpret_cast<long long*>(dist_cell(lb - 2, e)), sortable(dist));
              else *dist_cell(lb - 2, e) = dist;
              if (capped) atomicOr(&s_flags[e], URGYM_STATUS_GJK_ITER);
            }
          }
        }
      }
      if (armed) pose_slot.p = (URGYM_LDS double*)&s_pose[0][0] + tid;
      return armed;
    };
    if (HAS_OBST && WITH_EPA) {
      if (lane == 0) __hip_atomic_fetch_add(&s_left, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // (its marks are all set)
#pragma unroll 1
      for (;;) {
        // read the count BEFORE looking for marks: a mark set by a wave that had left by then is visible to the pass below
        const int left = __hip_atomic_load(&s_left, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool served = epa_pass();
        if (!served && left >= WAVES) break;
        if (!served) __builtin_amdgcn_s_sleep(16);
      }
    }
    __syncthreads();
  }
  STAMP_TIME(6);

  // ---- P4: one lane per env re-derives the end-effector frame (link 6 == ee_link 7, urdf:294-298) and finishes the step
  const int pe = (wv - (WAVES - G)) * GROUP + lane;  // env slot of this lane in P4 (waves WAVES-G .. WAVES-1)
  if (wv >= WAVES - G && pe < E && s_env[pe] != -1) {
    const int n = s_env[pe] >= 0 ? s_env[pe] : -2 - s_env[pe];
    double q[6];
    for (int i = 0; i < 6; i++) q[i] = LDS_Q ? s_q[i][pe] : joint_of_step<MODE>(P, actions, n, i);
    const bool ee_ready = (MODE == MODE_STEP) && (s_flags[pe] & EE_BIT);
    X3 TE = identity_x3();
    if (!ee_ready) {
      if (KIND == URGYM_ENV_ORI) {
        // UR5OriReach-v1 with the collision checks off (BASELINE configs[1], "FK + pose-distance reward only") spends its step HERE, one
        // lane per env: unrolled, the six sin / cos evaluations -- which do not depend on each other, only the chain products do -- can
        // be interleaved (+5.5 % at N = 4096).  The obstacle kernels keep the rolled loop: they reach this code only in their own
        // collision-free variant, and its registers are better spent on the GJK loop.
        double sn[6], cs[6];
#pragma unroll
        for (int k = 0; k < 6; k++) sincos(q[k], &sn[k], &cs[k]);
#pragma unroll
        for (int k = 0; k < 6; k++) fk_joint(TE, k, sn[k], cs[k]);
      } else {
#pragma unroll 1
        for (int k = 0; k < 6; k++) {
          double sn, cs;
          sincos(q[k], &sn, &cs);
          fk_joint(TE, k, sn, cs);
        }
      }
    }
    double opos[3] = {0, 0, 0};
    Q4 oq{0, 0, 0, 1};
    if (HAS_OBST) {
      if (LDS_STATE) {
        opos[0] = s_obst[0][pe]; opos[1] = s_obst[1][pe]; opos[2] = s_obst[2][pe];
        oq = Q4{s_obst[3][pe], s_obst[4][pe], s_obst[5][pe], s_obst[6][pe]};
      } else if (s_env[pe] >= 0 && P.sc_scratch != nullptr) {  // the P1 lane of this env (this very lane) cached the advanced pose
        for (int i = 0; i < 3; i++) opos[i] = SOA(P.sc_scratch, 12 + i, n, N);
        oq = Q4{SOA(P.sc_scratch, 15, n, N), SOA(P.sc_scratch, 16, n, N), SOA(P.sc_scratch, 17, n, N), SOA(P.sc_scratch, 18, n, N)};
      } else {
        obstacle_of_step<KIND>(P, n, opos, oq);
      }
    }
    const int step_count = B.step_count[n];
    float* row = &s_out[pe * 47];
    float ach[6];
    if (ee_ready) {
      for (int i = 0; i < 6; i++) ach[i] = B.observation[(size_t)n * OD + i];
    } else {
      Q4 eq = rot_to_quat(TE.r);
      double er, ep, ey;
      rpy_from_quat(eq, er, ep, ey);
      ach[0] = (float)TE.t.x; ach[1] = (float)TE.t.y; ach[2] = (float)TE.t.z;
      ach[3] = (float)er; ach[4] = (float)ep; ach[5] = (float)ey;
    }
    double goal[6];
    for (int i = 0; i < 6; i++) goal[i] = SOA(B.goal, i, n, N);
    double ld_old[5] = {0, 0, 0, 0, 0}, ld_new[5] = {0, 0, 0, 0, 0};
    bool coll = (s_flags[pe] & COLL_BIT) != 0;
    if (HAS_OBST) {
      for (int i = 0; i < 5; i++) {
        ld_new[i] = workbench ? unsortable(__double_as_longlong(*dist_cell(i, pe))) : *dist_cell(i, pe);
        if (MODE == MODE_STEP) ld_old[i] = SOA(B.link_dist, i, n, N);
        if (cfg.check_collision && ld_new[i] <= cfg.collision_margin) coll = true;
      }
    }
    if (MODE == MODE_PREFETCH) {
      // the rest of the record: what RESET would have written into the live state (reach.py:324-325, 680-681; set_velocity)
      const int key = s_key[pe], sl = key & 1;
      int rflags = s_flags[pe] & ~(COLL_BIT | EE_BIT | EPA_MASK);
      if (coll) rflags |= URGYM_STATUS_RESET_COLLISION;
      if (HAS_OBST) {
        for (int i = 0; i < 5; i++) REC(P, sl, REC_LD + i, n) = ld_new[i];
        REC(P, sl, REC_QUAT + 0, n) = oq.x; REC(P, sl, REC_QUAT + 1, n) = oq.y; REC(P, sl, REC_QUAT + 2, n) = oq.z; REC(P, sl, REC_QUAT + 3, n) = oq.w;
        double vel[6] = {0, 0, 0, 0, 0, 0};
        if (KIND == URGYM_ENV_DYN) {
          double st[6], en[6];
          for (int i = 0; i < 6; i++) { st[i] = REC(P, sl, REC_START + i, n); en[i] = REC(P, sl, REC_END + i, n); }
          dyn_velocity(st, en, cfg.dyn_time_duration, vel);
        }
        double dp[3];
        step_displacement(vel, cfg.dt, dp);
        for (int i = 0; i < 6; i++) REC(P, sl, REC_VEL + i, n) = vel[i];
        for (int i = 0; i < 3; i++) REC(P, sl, REC_DP + i, n) = dp[i];
      }
      RECI(P, sl, 1, n) = rflags;
      __threadfence();
      RECI(P, sl, 0, n) = key;  // the record is complete
    }
    if (MODE != MODE_PREFETCH) {
    // velocity slot of the Dyn observation (reach.py:657): STEP -> the velocity applied in this step; RESET/REFRESH ->
    // the stale ReachDyn.velocity of the previous step, which still sits in the observation buffer
    double vobs[6] = {0, 0, 0, 0, 0, 0};
    if (KIND == URGYM_ENV_DYN) {
      if (MODE == MODE_STEP) {
        if (step_count < cfg.dyn_motion_steps)
          for (int i = 0; i < 6; i++) vobs[i] = SOA(B.obst_vel, i, n, N);
      } else {
        for (int i = 0; i < 6; i++) vobs[i] = (double)B.observation[(size_t)n * OD + 24 + i];
      }
    }
    if (MODE == MODE_RESET && P.copy_final) {
      for (int i = 0; i < OD; i++) B.final_observation[(size_t)n * OD + i] = B.observation[(size_t)n * OD + i];
      for (int i = 0; i < GD; i++) {
        B.final_achieved_goal[(size_t)n * GD + i] = B.achieved_goal[(size_t)n * GD + i];
        B.final_desired_goal[(size_t)n * GD + i] = B.desired_goal[(size_t)n * GD + i];
      }
    }
    // observation row (core.py:252-261; UR5.py:320-325; reach.py:189, 307-308, 653-657)
    const double* ld_obs = (MODE == MODE_STEP) ? ld_old : ld_new;  // step(): link_dist lags one step (core.py:311 vs 316)
    auto write_row = [&](const float* ach_, const double* q_, const double* goal_, const double* obst6_, const double* opos_, Q4 oq_,
                         const double* vobs_, const double* ld_) {
      int p = 0;
      for (int i = 0; i < 6; i++) row[p++] = ach_[i];
      for (int i = 0; i < 6; i++) row[p++] = (float)q_[i];
      if (KIND == URGYM_ENV_ORI) {
        for (int i = 0; i < 6; i++) row[p++] = (float)goal_[i];
      } else if (KIND == URGYM_ENV_OBS) {
        for (int i = 0; i < 3; i++) row[p++] = (float)goal_[i];
        for (int i = 0; i < 6; i++) row[p++] = (float)obst6_[i];
        for (int i = 0; i < 5; i++) row[p++] = (float)ld_[i];
      } else {
        for (int i = 0; i < 6; i++) row[p++] = (float)goal_[i];
        for (int i = 0; i < 3; i++) row[p++] = (float)opos_[i];
        double r_, p_, y_;
        rpy_from_quat(oq_, r_, p_, y_);
        row[p++] = (float)r_; row[p++] = (float)p_; row[p++] = (float)y_;
        if (KIND == URGYM_ENV_DYN)  // ReachSta.get_obs has no velocity slot (reach.py:453-457)
          for (int i = 0; i < 6; i++) row[p++] = (float)vobs_[i];
        for (int i = 0; i < 5; i++) row[p++] = (float)ld_[i];
      }
      for (int i = 0; i < GD; i++) { row[OD + i] = ach_[i]; row[OD + GD + i] = (float)goal_[i]; }
    };
    double obst6[6] = {0, 0, 0, 0, 0, 0};
    if (KIND == URGYM_ENV_OBS)
      for (int i = 0; i < 6; i++) obst6[i] = SOA(B.obst_start, i, n, N);
    write_row(ach, q, goal, obst6, opos, oq, vobs, ld_obs);

    // is_success on the float32 achieved goal vs the float64 goal (reach.py:212-215, 348-350, 755-758)
    const double a6[6] = {(double)ach[0], (double)ach[1], (double)ach[2], (double)ach[3], (double)ach[4], (double)ach[5]};
    const double d = pos_distance(a6, goal);
    double th = 0.0;
    bool succ;
    if (KIND == URGYM_ENV_OBS) {
      succ = d < cfg.distance_threshold;
    } else {
      th = angular_distance(a6 + 3, goal + 3);
      succ = (d < cfg.distance_threshold) && (th < cfg.ori_threshold);
    }
    int flags = s_flags[pe] & ~(COLL_BIT | EE_BIT | EPA_MASK);
    if (MODE == MODE_STEP) {
      bool terminated = succ || coll;                 // core.py:313
      bool info_success = terminated ? !coll : false; // core.py:315
      double reward = 0.0;
      bool update_ld = false;
      if (KIND == URGYM_ENV_ORI) {  // reach.py:221-236
        reward += succ ? cfg.w_success : 0.0;
        reward += d * cfg.w_distance;
        reward += th * cfg.w_orientation;
        reward += coll ? cfg.w_collision : 0.0;
      } else if (KIND == URGYM_ENV_OBS) {  // reach.py:356-374
        reward += succ ? cfg.w_success : 0.0;
        reward += coll ? cfg.w_collision : 0.0;
        reward += cfg.w_distance * d;
        double sum = 0.0;
        for (int i = 0; i < 5; i++) sum += (ld_new[i] < cfg.near_threshold) ? cfg.w_link[i] * (ld_new[i] - ld_old[i]) : 0.0;
        reward += sum;
        update_ld = true;
      } else {  // reach.py:764-785
        if (coll) reward = cfg.w_collision;
        else if (succ) reward = cfg.w_success;
        else {
          reward += cfg.w_distance * d;
          reward += cfg.w_orientation * th;
          double sum = 0.0;
          for (int i = 0; i < 5; i++) sum += (ld_new[i] < cfg.near_threshold) ? cfg.w_link[i] * (ld_new[i] - ld_old[i]) : 0.0;
          reward += sum;
          update_ld = true;
        }
      }
      if (reward != reward) flags |= URGYM_STATUS_NAN;
      if (!update_ld) flags &= ~(URGYM_STATUS_PENETRATION | URGYM_STATUS_GJK_ITER);  // the distances were not consumed (reach.py:766-770)
      int sc = step_count + 1;
      bool truncated = sc >= cfg.max_episode_steps;  // TimeLimit (UR_gym/__init__.py:41)
      for (int i = 0; i < 6; i++) SOA(B.q, i, n, N) = q[i];
      B.step_count[n] = sc;
      if (KIND == URGYM_ENV_DYN || KIND == URGYM_ENV_STA) {
        for (int i = 0; i < 3; i++) SOA(B.obst_pos, i, n, N) = opos[i];
        SOA(B.obst_quat, 0, n, N) = oq.x; SOA(B.obst_quat, 1, n, N) = oq.y; SOA(B.obst_quat, 2, n, N) = oq.z; SOA(B.obst_quat, 3, n, N) = oq.w;
      }
      if (update_ld)
        for (int i = 0; i < 5; i++) SOA(B.link_dist, i, n, N) = ld_new[i];
      B.reward[n] = (float)reward;
      B.terminated[n] = terminated ? 1 : 0;
      B.truncated[n] = truncated ? 1 : 0;
      B.is_success[n] = info_success ? 1 : 0;
      B.collision[n] = coll ? 1 : 0;
      if (cfg.auto_reset && (terminated || truncated)) {
        bool consumed = false;
        if (KIND == URGYM_ENV_ORI && P.inline_ori) {
          // ---- ReachOri.reset (reach.py:197-200) is a goal draw that is never rejected: done here, what env_kernel<RESET> would do
          consumed = true;
          for (int i = 0; i < OD; i++) B.final_observation[(size_t)n * OD + i] = row[i];
          for (int i = 0; i < GD; i++) {
            B.final_achieved_goal[(size_t)n * GD + i] = row[OD + i];
            B.final_desired_goal[(size_t)n * GD + i] = row[OD + GD + i];
          }
          const int ecur = B.episode_id[n];
          double g2[6] = {0, 0, 0, 0, 0, 0}, none[6] = {0, 0, 0, 0, 0, 0}, q2[6], nold[5] = {0, 0, 0, 0, 0}, nop[3] = {0, 0, 0};
          sample_attempt<URGYM_ENV_ORI>(P, pose_slot, n, (uint32_t)ecur, 0, g2, none, none);
          for (int i = 0; i < 6; i++) { q2[i] = cfg.neutral_q[i]; SOA(B.goal, i, n, N) = g2[i]; SOA(B.q, i, n, N) = q2[i]; }
          B.step_count[n] = 0;
          B.episode_id[n] = ecur + 1;
          float ach2[6];
          for (int i = 0; i < 6; i++) ach2[i] = P.neutral_ach[i];
          write_row(ach2, q2, g2, none, nop, Q4{0, 0, 0, 1}, none, nold);
        } else if (P.prefetch) {
          // ---- inline auto-reset from the prefetched record of the env's next episode (valid iff it was computed for
          // exactly this episode id): what env_kernel<RESET> would do, minus the search.
          const int ecur = B.episode_id[n], sl = ecur & 1;
          if (RECI(P, sl, 0, n) == ecur) {
            consumed = true;
            // the step's row is the terminal observation
            for (int i = 0; i < OD; i++) B.final_observation[(size_t)n * OD + i] = row[i];
            for (int i = 0; i < GD; i++) {
              B.final_achieved_goal[(size_t)n * GD + i] = row[OD + i];
              B.final_desired_goal[(size_t)n * GD + i] = row[OD + GD + i];
            }
            double g2[6], st2[6], ld2[5] = {0, 0, 0, 0, 0}, v2[6] = {0, 0, 0, 0, 0, 0}, q2[6], op2[3] = {0, 0, 0};
            Q4 oq2{0, 0, 0, 1};
            for (int i = 0; i < 6; i++) { g2[i] = REC(P, sl, REC_GOAL + i, n); st2[i] = REC(P, sl, REC_START + i, n); q2[i] = cfg.neutral_q[i]; }
            for (int i = 0; i < 6; i++) { SOA(B.goal, i, n, N) = g2[i]; SOA(B.q, i, n, N) = q2[i]; }
            if (HAS_OBST) {
              for (int i = 0; i < 6; i++) {
                SOA(B.obst_start, i, n, N) = st2[i];
                SOA(B.obst_end, i, n, N) = REC(P, sl, REC_END + i, n);
                SOA(B.obst_vel, i, n, N) = REC(P, sl, REC_VEL + i, n);
              }
              for (int i = 0; i < 3; i++) SOA(B.obst_vel, 6 + i, n, N) = REC(P, sl, REC_DP + i, n);
              oq2 = Q4{REC(P, sl, REC_QUAT + 0, n), REC(P, sl, REC_QUAT + 1, n), REC(P, sl, REC_QUAT + 2, n), REC(P, sl, REC_QUAT + 3, n)};
              for (int i = 0; i < 3; i++) { op2[i] = st2[i]; SOA(B.obst_pos, i, n, N) = st2[i]; }
              SOA(B.obst_quat, 0, n, N) = oq2.x; SOA(B.obst_quat, 1, n, N) = oq2.y; SOA(B.obst_quat, 2, n, N) = oq2.z; SOA(B.obst_quat, 3, n, N) = oq2.w;
              for (int i = 0; i < 5; i++) { ld2[i] = REC(P, sl, REC_LD + i, n); SOA(B.link_dist, i, n, N) = ld2[i]; }
              // Dyn: the stale ReachDyn.velocity in the reset observation is the float32 value of this step's row
              if (KIND == URGYM_ENV_DYN)
                for (int i = 0; i < 6; i++) v2[i] = (double)row[24 + i];
            }
            B.step_count[n] = 0;
            B.episode_id[n] = ecur + 1;
            flags |= RECI(P, sl, 1, n);
            // reset observation: the neutral pose's end-effector frame (a constant, evaluated once at urgym_create by the
            // same device code the RESET kernel runs)
            float ach2[6];
            for (int i = 0; i < 6; i++) ach2[i] = P.neutral_ach[i];
            write_row(ach2, q2, g2, st2, op2, oq2, v2, ld2);
            // the slot is free again: its next occupant is the episode after next
            const int pos = atomicAdd(P.rcount, 1);
            if (pos < P.rcap) P.rlist[pos] = make_int2(n, ecur + 2);
          }
        }
        if (!consumed) {
          int slot = atomicAdd(&B.done_count[P.pp], 1);
          B.done_list[slot] = n;
          // no fallback launch follows (the host believed every record valid): the env stays un-reset and says so
          if (P.prefetch && !P.fallback_on) atomicOr(&B.status[n], URGYM_STATUS_STALE_RECORD);
        }
      }
    } else {
      // reset / refresh: fresh link_dist == last_dist (reach.py:324-325, 680-681, 711-713), obstacle pose + velocity
      if (HAS_OBST) {
        for (int i = 0; i < 3; i++) SOA(B.obst_pos, i, n, N) = opos[i];
        SOA(B.obst_quat, 0, n, N) = oq.x; SOA(B.obst_quat, 1, n, N) = oq.y; SOA(B.obst_quat, 2, n, N) = oq.z; SOA(B.obst_quat, 3, n, N) = oq.w;
        for (int i = 0; i < 5; i++) SOA(B.link_dist, i, n, N) = ld_new[i];
        double vel[6] = {0, 0, 0, 0, 0, 0};
        if (KIND == URGYM_ENV_DYN) {
          double st[6], en[6];
          for (int i = 0; i < 6; i++) { st[i] = SOA(B.obst_start, i, n, N); en[i] = SOA(B.obst_end, i, n, N); }
          dyn_velocity(st, en, cfg.dyn_time_duration, vel);
        }
        double dp[3];
        step_displacement(vel, cfg.dt, dp);
        for (int i = 0; i < 6; i++) SOA(B.obst_vel, i, n, N) = vel[i];
        for (int i = 0; i < 3; i++) SOA(B.obst_vel, 6 + i, n, N) = dp[i];
      }
      if (MODE == MODE_RESET) {
        B.step_count[n] = 0;
        if (coll) flags |= URGYM_STATUS_RESET_COLLISION;
        if (P.prefetch) {  // records of the next two episodes of this env (refilled by the PREFETCH launch that follows)
          const int k0 = s_key[pe];
          const int pos = atomicAdd(P.rcount, 2);
          if (pos + 1 < P.rcap) { P.rlist[pos] = make_int2(n, k0); P.rlist[pos + 1] = make_int2(n, k0 + 1); }
        }
        if (!P.copy_final) {
          B.reward[n] = 0.f; B.terminated[n] = 0; B.truncated[n] = 0;
          B.is_success[n] = succ ? 1 : 0;
          B.collision[n] = coll ? 1 : 0;
        }
      } else {
        B.is_success[n] = succ ? 1 : 0;
        B.collision[n] = coll ? 1 : 0;
      }
    }
    if (flags) atomicOr(&B.status[n], flags);
    }  // MODE != MODE_PREFETCH
  }
  STAMP_TIME(7);
  STAMP(9, __builtin_amdgcn_s_memrealtime());
  __syncthreads();

  // ---- write-back of the observation rows staged in LDS (coalesced for STEP: the group's rows are contiguous)
  if (MODE == MODE_STEP) {
    const int base = first;
    const int cnt = min(E, N - base);
    // (streaming stores: nothing in this launch reads the rows again, and they should not push the hull tables out of L2)
    for (int i = tid; i < cnt * OD; i += THREADS) __builtin_nontemporal_store(s_out[(i / OD) * 47 + (i % OD)], &B.observation[(size_t)base * OD + i]);
    for (int i = tid; i < cnt * GD; i += THREADS) {
      __builtin_nontemporal_store(s_out[(i / GD) * 47 + OD + (i % GD)], &B.achieved_goal[(size_t)base * GD + i]);
      __builtin_nontemporal_store(s_out[(i / GD) * 47 + OD + GD + (i % GD)], &B.desired_goal[(size_t)base * GD + i]);
    }
    if (bidx == 0 && tid == 0) B.done_count[P.pp ^ 1] = 0;  // arm the other counter
  } else if (MODE != MODE_PREFETCH) {
    const int cnt = min(E, list_count - first);
    for (int i = tid; i < cnt * OD; i += THREADS) {
      const int e = i / OD;
      const int se = s_env[e];
      const int ne = se >= 0 ? se : -2 - se;
      B.observation[(size_t)ne * OD + (i % OD)] = s_out[e * 47 + (i % OD)];
    }
    for (int i = tid; i < cnt * GD; i += THREADS) {
      const int e = i / GD;
      const int se = s_env[e];
      const int ne = se >= 0 ? se : -2 - se;
      B.achieved_goal[(size_t)ne * GD + (i % GD)] = s_out[e * 47 + OD + (i % GD)];
      B.desired_goal[(size_t)ne * GD + (i % GD)] = s_out[e * 47 + OD + GD + (i % GD)];
    }
  }
}

template <int KIND, int MODE, bool WITH_EPA>
__global__ void __launch_bounds__(THREADS, ((MODE == MODE_STEP || MODE == MODE_PREFETCH) ? URGYM_RESIDENT : 2)) env_kernel(const KParams P, const float* __restrict__ actions) {
  __shared__ EnvLds<MODE> lds;
  env_body<KIND, MODE, WITH_EPA>(P, actions, lds, (int)blockIdx.x, (int)gridDim.x);
}

// The steady-state step launch: `step_blocks` STEP workgroups (parameters Ps) followed by the PREFETCH workgroups (Pr) that refill
// the episode records consumed by the PREVIOUS step.  One launch instead of a step kernel on the caller's stream plus a refill
// kernel on a side stream with two events and two stream waits per step; the refill workgroups come last in the grid, so they take
// the slots the step workgroups free in the tail of the launch.  The two kinds never touch the same data (the refill writes record
// slots the concurrent step cannot read -- episode parity -- and reads nothing the step writes: the episode id travels in its list).
template <int KIND, bool WITH_EPA>
__global__ void __launch_bounds__(THREADS, URGYM_RESIDENT) env_step_fused(const KParams Ps, const KParams Pr, const float* __restrict__ actions, const int step_blocks) {
  __shared__ union FusedLds {
    EnvLds<MODE_STEP> step;
    EnvLds<MODE_PREFETCH> refill;
  } lds;
  if ((int)blockIdx.x < step_blocks) {
    env_body<KIND, MODE_STEP, WITH_EPA>(Ps, actions, lds.step, (int)blockIdx.x, step_blocks);
  } else {
    // the refill workgroups stride over the list in chunks of PREFETCH_MAX_ENVS entries: their number is a scheduling choice of the
    // host (launch_fused), any number serves any list length
    const int rb = (int)blockIdx.x - step_blocks, nrb = (int)gridDim.x - step_blocks;
    const int count = min(*Pr.rcount, Pr.rcap);  // complete: the launch that appended to this list has finished
#pragma unroll 1
    for (int chunk = rb; chunk * PREFETCH_MAX_ENVS < count; chunk += nrb) {
      env_body<KIND, MODE_PREFETCH, true>(Pr, nullptr, lds.refill, chunk, nrb);
      __syncthreads();  // the next chunk re-uses the LDS slots
    }
  }
}

// end-effector pose of a joint vector exactly as P4 derives it (urgym_create evaluates the neutral pose once)
__global__ void ee_pose_kernel(const double* q, float* out) {
  X3 TE = identity_x3();
  for (int k = 0; k < 6; k++) {
    double sn, cs;
    sincos(q[k], &sn, &cs);
    fk_joint(TE, k, sn, cs);
  }
  double er, ep, ey;
  rpy_from_quat(rot_to_quat(TE.r), er, ep, ey);
  out[0] = (float)TE.t.x; out[1] = (float)TE.t.y; out[2] = (float)TE.t.z;
  out[3] = (float)er; out[4] = (float)ep; out[5] = (float)ey;
}

// unit probe: one closest-distance query per lane through the very same device GJK (tests only; not on the hot path)
__global__ void probe_closest_kernel(HullMap g, int count, const int* type_a, const double* par_a, const double* pose_a,
                                     const int* type_b, const double* par_b, const double* pose_b, double threshold,
                                     double* out_dist, int* out_info) {
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i0 < count;
  const int i = live ? i0 : count - 1;  // idle lanes shadow the last query (the wave stays whole for the EPA below)
  auto mk = [](int type, const double* par, double& margin) {
    ShapeDesc s;
    s.type = type; s.hull = 0; s.hx = s.hy = s.hz = 0.0;
    if (type == SH_HULL) { s.hull = (int)par[0] - 1; margin = M_HULL; }
    else if (type == SH_CYLZ) { const double m = M_PRIM; s.hx = s.hy = par[0] - m; s.hz = 0.5 * par[1] - m; margin = m; }
    else if (type == SH_BOX) { const double m = M_PRIM; s.hx = par[0] - m; s.hy = par[1] - m; s.hz = par[2] - m; margin = m; }
    else { margin = par[0]; }
    return s;
  };
  double ma, mb;
  ShapeDesc A = mk(type_a[i], par_a + 3 * i, ma), Bs = mk(type_b[i], par_b + 3 * i, mb);
  X3 Ta, Tb;
  const double* pa = pose_a + 7 * i;
  const double* pb = pose_b + 7 * i;
  quat_to_rot(Q4{pa[3], pa[4], pa[5], pa[6]}, Ta.r);
  Ta.t = d3(pa[0], pa[1], pa[2]);
  quat_to_rot(Q4{pb[3], pb[4], pb[5], pb[6]}, Tb.r);
  Tb.t = d3(pb[0], pb[1], pb[2]);
  __shared__ double s_pose[GJK_SLOT_DOUBLES][64];
  const XRef slot{(URGYM_LDS double*)&s_pose[0][0] + threadIdx.x, 64};
  int info = 0;
  double core = 0.0;
  if (live) {
    store(slot, rel(Tb, Ta));
    core = gjk_core_distance(g, A, slot, Bs, rotT(Tb, d3(0, 1, 0)), ma + mb + 0.02 + threshold, info);
    out_dist[i] = core - ma - mb;
    out_info[i] = info;
  }
  // overlapping cores: penetration depth by the wave-cooperative EPA, query by query
  unsigned long long pen = __ballot(live && (info & GJK_PENETRATING));
  const EpaWs ws{(URGYM_LDS double*)&s_pose[0][0], 64};
#pragma unroll 1
  while (pen) {
    const int l = __builtin_ctzll(pen);
    pen &= pen - 1ull;
    const int j = blockIdx.x * blockDim.x + l;  // every lane rebuilds the operands of query j
    double mja, mjb;
    const ShapeDesc Aj = mk(type_a[j], par_a + 3 * j, mja), Bj = mk(type_b[j], par_b + 3 * j, mjb);
    X3 Tja, Tjb;
    const double* qa = pose_a + 7 * j;
    const double* qb = pose_b + 7 * j;
    quat_to_rot(Q4{qa[3], qa[4], qa[5], qa[6]}, Tja.r);
    Tja.t = d3(qa[0], qa[1], qa[2]);
    quat_to_rot(Q4{qb[3], qb[4], qb[5], qb[6]}, Tjb.r);
    Tjb.t = d3(qb[0], qb[1], qb[2]);
    epa_wave_sync();
    store(XRef{ws.base, ws.stride}, rel(Tjb, Tja));
    epa_wave_sync();
    bool capped;
    const double depth = epa_wave(g, Aj, Bj, ws, (int)threadIdx.x, capped);
    const int info_l = __shfl(info, l);
    if ((int)threadIdx.x == l) {  // (the lane that owns query j: its earlier provisional stores are overwritten in order)
      out_dist[j] = -(depth + mja + mjb);
      out_info[j] = info_l | (capped ? GJK_ITERCAP : 0);
    }
    epa_wave_sync();
  }
}

// unit probe: utils.distance / utils.angular_distance through the very device functions P4 calls (tests only)
__global__ void probe_pose_distance_kernel(int count, const double* a6, const double* b6, double* out2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double a[6], b[6];
  for (int k = 0; k < 6; k++) { a[k] = a6[6 * i + k]; b[k] = b6[6 * i + k]; }
  out2[2 * i + 0] = pos_distance(a, b);
  out2[2 * i + 1] = angular_distance(a + 3, b + 3);
}

// rows 6..8 of obst_vel (displacement of one env step) re-derived from the twist in rows 0..5, for every env: what reset / refresh
// do after they have computed the twist, for a caller that wrote a twist of its own (urgym_derive_obstacle_motion)
__global__ void derive_displacement_kernel(double* obst_vel, int N, double dt) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double vel[6], dp[3];
  for (int i = 0; i < 6; i++) vel[i] = obst_vel[(size_t)i * N + n];
  step_displacement(vel, dt, dp);
  for (int i = 0; i < 3; i++) obst_vel[(size_t)(6 + i) * N + n] = dp[i];
}

// compaction of an explicit reset / refresh mask into done_list (mask == nullptr: every env)
__global__ void build_list_kernel(const uint8_t* mask, int N, int* list, int* count) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (mask == nullptr) {
    list[i] = i;
    if (i == 0) *count = N;
  } else if (mask[i]) {
    int slot = atomicAdd(count, 1);
    list[slot] = i;
  }
}

// ------------------------------------------------------------------------------------------------- host side
struct Handle {
  urgym_config cfg;
  urgym_buffers buf;
  bool bound = false;
  int device = 0;
  int obs_dim = 0, goal_dim = 0;
  double* d_ld_scratch = nullptr;  // [5][N] link distances of the running step
  double* d_sc_scratch = nullptr;  // [SC_FIELDS][N] set-up cache of the running step
  bool sc_frames = false;          // ... with the link frames (URGYM_SETUP_CACHE=2; default 1: without them; 0: no cache at all)
  CandRec* d_recs = nullptr;          // support map: candidate records ...
  unsigned short* d_cell = nullptr;   // ... and the cube map of directions that points into them
  uint64_t seed = 0;
  int pp = 0;
  int step_envs = GROUP;  // envs per workgroup of the step kernel (see urgym_create)
  int big_blocks = 0, tail_envs = 0;  // two-tier geometry of the step launch: the first big_blocks workgroups serve step_envs, the rest tail_envs
  int reset_envs = 4;   // envs per workgroup of the auto-reset kernel (latency-bound: few envs, spread wide)
  char err[512] = {0};
  bool launch_refused = false;  // a launch was not issued because its geometry failed validation (err says why)
  // timing
  bool timing = false;
  int timing_every = 1;   // time every k-th step (an event pair costs the stream ~6 us: sampling keeps the measurement out of the measured)
  long timing_tick = 0;
  std::vector<hipEvent_t> ev;  // pairs: [2i] start, [2i+1] stop ; kind in ev_kind
  std::vector<int> ev_kind;    // 0 = step kernel, 1 = reset kernel(s) on the caller's stream, 2 = overlapped refill
  size_t ev_used = 0;
  double last_refill_us = 0.0;
  // prefetched episode records (DESIGN.md "auto-reset off the critical path")
  bool prefetch = false;
  bool inline_ori = false;  // UR5OriReach-v1: finished envs are reset inside the step kernel (no RESET launch per step)
  float neutral_ach[6] = {0, 0, 0, 0, 0, 0};
  double* d_rec = nullptr;      // [2][REC_FIELDS][N]
  int32_t* d_reci = nullptr;    // [2][2][N]
  int2* d_rl[4] = {nullptr, nullptr, nullptr, nullptr};  // refill lists: three rotating asynchronous ones, one synchronous (index 3)
  int rl_cap[4] = {0, 0, 0, 0};
  int* d_rcount = nullptr;      // their counters
  int parity = 0;
  uint64_t rec_seed = 0;
  bool rec_seed_valid = false;
  // Steps left in which a finished env may still meet a record that is not valid for its episode (after create / bind /
  // urgym_invalidate_records / a reset that did not cover every env): only then does a step carry the fallback launches.
  // An env that falls back gets fresh records for its next two episodes, and every env finishes within max_episode_steps.
  int dirty_steps = 0;
  int refill_blocks_override = 0;   // URGYM_REFILL_BLOCKS (tuning / tests): refill workgroups per fused launch, 0 = the policy of launch_fused
  long steps_since_full_reset = -1; // step launches since the last urgym_reset of every env (-1: none yet)
  // the refill of a step's finished envs rides in the NEXT launch: a burst that happens in step k * max_episode_steps (counted from 1)
  // is served one launch later; allow a step of slack on either side
  bool burst_due() const {
    if (steps_since_full_reset < 0 || cfg.max_episode_steps < 4) return true;
    const long r = steps_since_full_reset % cfg.max_episode_steps;
    return steps_since_full_reset >= cfg.max_episode_steps - 1 && (r <= 2 || r == cfg.max_episode_steps - 1);
  }
};
thread_local char g_err[512] = {0};

int fail(Handle* h, int code, const char* what, hipError_t e = hipSuccess) {
  char* dst = h ? h->err : g_err;
  if (e != hipSuccess)
    snprintf(dst, 512, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(dst, 512, "%s", what);
  return code;
}
#define HIP_TRY(h, call)                                              \
  do {                                                                \
    hipError_t _e = (call);                                           \
    if (_e != hipSuccess) return fail(h, URGYM_ERR_HIP, #call, _e);   \
  } while (0)

void fill_default(int env_kind, int num_envs, urgym_config* c) {
  memset(c, 0, sizeof(*c));
  c->env_kind = env_kind;
  c->num_envs = num_envs;
  c->max_episode_steps = 100;  // UR_gym/__init__.py:41
  c->auto_reset = 1;
  c->check_collision = 1;
  c->max_reset_tries = 4096;
  c->gjk_start = URGYM_GJK_START_BULLET;
  c->dyn_motion_steps = 25;    // reach.py:735
  c->action_scale = M_PI * 0.1;
  c->dt = 20.0 / 500.0;        // pyb_setup.py:25,40
  c->distance_threshold = 0.05;
  c->ori_threshold = 0.0873;
  c->w_collision = -500;
  c->w_success = 200;
  c->near_threshold = 0.2;
  c->collision_margin = 0.01;
  c->target_clearance = 0.1;
  c->min_travel = 1.0;
  c->dyn_time_duration = 2.0;
  const double neutral[6] = {0.0, -1.5708, 0.0, -1.5708, 0.0, 0.0};  // UR5.py:262
  for (int i = 0; i < 6; i++) c->neutral_q[i] = neutral[i];
  if (env_kind == URGYM_ENV_ORI) {  // reach.py:148-157
    c->w_distance = -70; c->w_orientation = -30;
    const double gl[3] = {0.3, -0.5, 0.0}, gh[3] = {0.75, 0.5, 0.2};
    for (int i = 0; i < 3; i++) { c->goal_low[i] = gl[i]; c->goal_high[i] = gh[i]; }
  } else if (env_kind == URGYM_ENV_OBS) {  // reach.py:246-256
    c->w_distance = -100; c->w_orientation = 0;
    const double gl[3] = {0.3, -0.5, -0.1}, gh[3] = {0.75, 0.5, 0.2}, ol[3] = {0.5, -0.5, 0.25}, oh[3] = {1.0, 0.5, 0.55};
    for (int i = 0; i < 3; i++) { c->goal_low[i] = gl[i]; c->goal_high[i] = gh[i]; c->obst_low[i] = ol[i]; c->obst_high[i] = oh[i]; }
    for (int i = 0; i < 5; i++) c->w_link[i] = 100.0;
  } else {  // Dyn: reach.py:584-598; Sta: reach.py:385-399 (Ori's goal box, Obs's obstacle box, Dyn's weights)
    c->w_distance = -70; c->w_orientation = -30;
    const double gl[3] = {0.4, -0.5, 0.0}, gh[3] = {0.75, 0.5, 0.2}, ol[3] = {0.5, -0.8, 0.25}, oh[3] = {1.2, 0.8, 0.75};
    const double sgl[3] = {0.3, -0.5, 0.0}, sgh[3] = {0.75, 0.5, 0.2}, sol[3] = {0.5, -0.5, 0.25}, soh[3] = {1.0, 0.5, 0.55};
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
}

KParams make_params(Handle* h, int copy_final) {
  KParams P;
  P.cfg = h->cfg;
  P.buf = h->buf;
  P.graph.recs = h->d_recs;
  P.graph.cell = h->d_cell;
  P.obs_dim = h->obs_dim;
  P.goal_dim = h->goal_dim;
  P.seed_lo = (uint32_t)h->seed;
  P.seed_hi = (uint32_t)(h->seed >> 32);
  P.pp = h->pp;
  P.copy_final = copy_final;
  P.envs = GROUP;
  P.big_blocks = 0;
  P.envs_tail = 0;
  P.rec_d = h->d_rec;
  P.rec_i = h->d_reci;
  P.rlist = nullptr;
  P.rcount = nullptr;
  P.rcap = 0;
  P.ld_scratch = h->d_ld_scratch;
  P.sc_scratch = h->d_sc_scratch;
  P.sc_frames = h->sc_frames ? 1 : 0;
  P.rzero = nullptr;
  P.rzero2 = nullptr;
  P.fallback_on = 1;
  P.prefetch = 0;
  P.inline_ori = h->inline_ori ? 1 : 0;
  for (int i = 0; i < 6; i++) P.neutral_ach[i] = h->neutral_ach[i];
  return P;
}

// A STEP grid must cover env 0 .. n-1 exactly once with workgroups of at most STEP_MAX_ENVS envs: the kernel derives every per-env
// index (LDS slots, rows of the scratch arrays, observation rows) from (workgroup index, P.envs, P.big_blocks, P.envs_tail).
bool step_geometry_ok(const KParams& P, long n, long blocks) {
  if (P.envs < 1 || P.envs > STEP_MAX_ENVS || blocks < 1) return false;
  if (P.envs_tail == 0) return (blocks - 1) * P.envs < n && blocks * P.envs >= n;
  if (P.envs_tail < 1 || P.envs_tail > STEP_MAX_ENVS || P.big_blocks < 1 || P.big_blocks >= blocks) return false;
  const long covered_before_last = (long)P.big_blocks * P.envs + (blocks - P.big_blocks - 1) * P.envs_tail;
  return covered_before_last < n && covered_before_last + P.envs_tail >= n;
}

template <int MODE>
void launch_mode(Handle* h, KParams P, const float* actions, int envs, hipStream_t s, long items = -1) {
  const int cap = (MODE == MODE_PREFETCH) ? PREFETCH_MAX_ENVS : ((MODE == MODE_STEP) ? STEP_MAX_ENVS : MAX_ENVS);
  envs = envs < 1 ? 1 : (envs > cap ? cap : envs);              // the kernel's LDS is sized for that many
  P.envs = envs;
  if (items < 0) items = h->cfg.num_envs;                       // list-driven launches: an upper bound of the list length
  long blocks = (items + envs - 1) / envs;
  if (MODE == MODE_STEP && h->tail_envs > 0 && h->big_blocks > 0 && (long)h->big_blocks * envs < items) {
    P.big_blocks = h->big_blocks;
    P.envs_tail = h->tail_envs;
    blocks = h->big_blocks + (items - (long)h->big_blocks * envs + h->tail_envs - 1) / h->tail_envs;
  }
  dim3 grid((unsigned)blocks), block(THREADS);
  if (MODE == MODE_STEP && !step_geometry_ok(P, items, blocks)) { snprintf(h->err, sizeof(h->err), "step launch geometry does not cover the envs exactly once"); h->launch_refused = true; return; }
  // which launches can consume a penetration depth: see need_epa in the kernel
  const bool epa = (MODE != MODE_STEP) || !h->cfg.check_collision;
  switch (h->cfg.env_kind) {
    case URGYM_ENV_ORI: hipLaunchKernelGGL((env_kernel<URGYM_ENV_ORI, MODE, false>), grid, block, 0, s, P, actions); break;
    case URGYM_ENV_OBS: hipLaunchKernelGGL((env_kernel<URGYM_ENV_OBS, MODE, true>), grid, block, 0, s, P, actions); break;
    case URGYM_ENV_STA:
      if (epa) hipLaunchKernelGGL((env_kernel<URGYM_ENV_STA, MODE, true>), grid, block, 0, s, P, actions);
      else hipLaunchKernelGGL((env_kernel<URGYM_ENV_STA, MODE, false>), grid, block, 0, s, P, actions);
      break;
    default:
      if (epa) hipLaunchKernelGGL((env_kernel<URGYM_ENV_DYN, MODE, true>), grid, block, 0, s, P, actions);
      else hipLaunchKernelGGL((env_kernel<URGYM_ENV_DYN, MODE, false>), grid, block, 0, s, P, actions);
      break;
  }
}

// The steady-state step of the obstacle envs: STEP workgroups + the PREFETCH workgroups that refill what the previous step consumed
// (env_step_fused).  Ps / Pr: parameters of the two parts.
void launch_fused(Handle* h, KParams Ps, KParams Pr, const float* actions, hipStream_t s) {
  const int envs = h->step_envs < 1 ? 1 : (h->step_envs > STEP_MAX_ENVS ? STEP_MAX_ENVS : h->step_envs);
  const long n = h->cfg.num_envs;
  Ps.envs = envs;
  long step_blocks = (n + envs - 1) / envs;
  if (h->tail_envs > 0 && h->big_blocks > 0 && (long)h->big_blocks * envs < n) {
    Ps.big_blocks = h->big_blocks;
    Ps.envs_tail = h->tail_envs;
    step_blocks = h->big_blocks + (n - (long)h->big_blocks * envs + h->tail_envs - 1) / h->tail_envs;
  }
  Pr.envs = PREFETCH_MAX_ENVS;
  // Refill workgroups: they stride over the list, so their number only decides how parallel the refill is.  About 1.6 % of the envs
  // finish per step under a random policy (N / 1920 chunks of 32); the grid carries four times that, at least 64 -- and the whole
  // list's worth (one workgroup per possible chunk) in the steps where a burst is due: every env that survives from a full reset is
  // truncated max_episode_steps later, all in the same step (and their successors again a period later).
  const long full = ((long)Pr.rcap + PREFETCH_MAX_ENVS - 1) / PREFETCH_MAX_ENVS;
  long refill_blocks = std::max(64L, 4 * (n / 1920 + 1));
  if (h->refill_blocks_override > 0) refill_blocks = h->refill_blocks_override;
  else if (h->burst_due()) refill_blocks = full;
  if (refill_blocks > full) refill_blocks = full;
  dim3 grid((unsigned)(step_blocks + refill_blocks)), block(THREADS);
  if (!step_geometry_ok(Ps, n, step_blocks)) { snprintf(h->err, sizeof(h->err), "step launch geometry does not cover the envs exactly once"); h->launch_refused = true; return; }
  const bool epa = !h->cfg.check_collision;  // (which STEP launches can consume a penetration depth: see need_epa in the kernel)
  const int sb = (int)step_blocks;
  switch (h->cfg.env_kind) {
    case URGYM_ENV_OBS: hipLaunchKernelGGL((env_step_fused<URGYM_ENV_OBS, true>), grid, block, 0, s, Ps, Pr, actions, sb); break;
    case URGYM_ENV_STA:
      if (epa) hipLaunchKernelGGL((env_step_fused<URGYM_ENV_STA, true>), grid, block, 0, s, Ps, Pr, actions, sb);
      else hipLaunchKernelGGL((env_step_fused<URGYM_ENV_STA, false>), grid, block, 0, s, Ps, Pr, actions, sb);
      break;
    default:
      if (epa) hipLaunchKernelGGL((env_step_fused<URGYM_ENV_DYN, true>), grid, block, 0, s, Ps, Pr, actions, sb);
      else hipLaunchKernelGGL((env_step_fused<URGYM_ENV_DYN, false>), grid, block, 0, s, Ps, Pr, actions, sb);
      break;
  }
}

int time_begin(Handle* h, int kind, hipStream_t s) {
  if (!h->timing || h->ev_used >= 65536) return -1;
  if (kind == 0 && (h->timing_tick++ % h->timing_every) != 0) return -1;
  if (h->ev_used + 2 > h->ev.size()) {
    for (int i = 0; i < 2; i++) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return -1;
      h->ev.push_back(e);
    }
    h->ev_kind.push_back(kind);
  }
  h->ev_kind[h->ev_used / 2] = kind;
  hipEventRecord(h->ev[h->ev_used], s);
  return (int)h->ev_used;
}
void time_end(Handle* h, int slot, hipStream_t s) {
  if (slot < 0) return;
  hipEventRecord(h->ev[slot + 1], s);
  h->ev_used = slot + 2;
}

int check_bound(Handle* h) {
  if (!h) return fail(nullptr, URGYM_ERR_ARG, "null handle");
  if (!h->bound) return fail(h, URGYM_ERR_STATE, "urgym_bind() has not been called");
  return URGYM_OK;
}

void release_prefetch(Handle* h) {
  if (h->d_rec) { hipFree(h->d_rec); h->d_rec = nullptr; }
  if (h->d_reci) { hipFree(h->d_reci); h->d_reci = nullptr; }
  for (int i = 0; i < 4; i++)
    if (h->d_rl[i]) { hipFree(h->d_rl[i]); h->d_rl[i] = nullptr; }
  if (h->d_rcount) { hipFree(h->d_rcount); h->d_rcount = nullptr; }
}

// list `which` (0 .. 2: asynchronous, 3: synchronous) as the refill list of a launch
void use_list(Handle* h, KParams& P, int which) {
  P.rlist = h->d_rl[which];
  P.rcount = h->d_rcount + which;
  P.rcap = h->rl_cap[which];
  P.prefetch = 1;
}

int do_step(Handle* h, const float* actions, hipStream_t s) {
  KParams P = make_params(h, 1);
  const bool pf = h->prefetch && h->cfg.auto_reset;
  int slot;
  if (!pf) {
    slot = time_begin(h, 0, s);
    launch_mode<MODE_STEP>(h, P, actions, h->step_envs, s);
    time_end(h, slot, s);
    if (h->cfg.auto_reset && !h->inline_ori) {
      slot = time_begin(h, 1, s);
      launch_mode<MODE_RESET>(h, P, nullptr, h->reset_envs, s);  // ~1 % of the envs per step: small workgroups, many CUs
      time_end(h, slot, s);
    }
  } else {
    // Finished envs are reset inline from their prefetched episode records, and the record slots a step consumes are refilled by
    // PREFETCH workgroups that ride in the NEXT step's launch.  Three refill lists rotate: the STEP part of launch t appends to list
    // t % 3, the PREFETCH part of launch t + 1 reads it, and the STEP part of launch t + 2 re-arms it (its reader belongs to a
    // launch that has completed by then) -- no extra launch, memset, event or second stream.  A record consumed at step t is
    // whole again when launch t + 1 ends, i.e. before step t + 2 could need it; at step t + 1 the env uses its other slot.
    const int cur = h->parity, nxt = (cur + 1) % 3, prv = (cur + 2) % 3;
    const bool dirty = h->dirty_steps > 0;
    use_list(h, P, cur);
    P.rzero = h->d_rcount + nxt;
    P.rzero2 = dirty ? h->d_rcount + 3 : nullptr;  // the synchronous list of this step's fallback launches
    P.fallback_on = dirty ? 1 : 0;
    KParams Pr = P;
    use_list(h, Pr, prv);
    Pr.rzero = Pr.rzero2 = nullptr;
    slot = time_begin(h, 0, s);
    launch_fused(h, P, Pr, actions, s);
    time_end(h, slot, s);
    // In steady state every record is valid (an env's two slots hold its next two episodes, and a consumed slot is refilled before
    // the env can need it again).  Only while records may be stale (h->dirty_steps > 0: first use, a new binding,
    // urgym_invalidate_records) does the step carry the fallback: the RESET kernel for envs that found no valid record, then
    // PREFETCH for their next two episodes -- unbounded, so that an env that fell back comes out clean.
    if (dirty) {
      slot = time_begin(h, 1, s);
      KParams Pf = P;
      use_list(h, Pf, 3);
      Pf.rzero = Pf.rzero2 = nullptr;
      launch_mode<MODE_RESET>(h, Pf, nullptr, GROUP, s);
      launch_mode<MODE_PREFETCH>(h, Pf, nullptr, 8, s, h->rl_cap[3]);
      time_end(h, slot, s);
      h->dirty_steps--;
    }
    h->parity = nxt;
  }
  if (h->launch_refused) { h->launch_refused = false; return URGYM_ERR_STATE; }
  if (h->steps_since_full_reset >= 0) h->steps_since_full_reset++;
  h->pp ^= 1;
  HIP_TRY(h, hipGetLastError());
  return URGYM_OK;
}

int do_masked(Handle* h, const uint8_t* mask, int mode, hipStream_t s) {
  const int N = h->cfg.num_envs;
  HIP_TRY(h, hipMemsetAsync(h->buf.done_count + h->pp, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(build_list_kernel, dim3((N + 255) / 256), dim3(256), 0, s, mask, N, h->buf.done_list, h->buf.done_count + h->pp);
  KParams P = make_params(h, 0);
  if (mode == MODE_RESET) {
    const bool pf = h->prefetch && h->cfg.auto_reset;
    if (pf) {
      if (!h->rec_seed_valid || h->rec_seed != h->seed) {  // records are keyed with the seed: a new one invalidates them all
        HIP_TRY(h, hipMemsetAsync(h->d_reci, 0xFF, sizeof(int32_t) * 4 * (size_t)N, s));
        h->rec_seed = h->seed;
        h->rec_seed_valid = true;
        h->dirty_steps = h->cfg.max_episode_steps + 1;
      } else if (mask != nullptr) {
        // the slots the last step consumed are refilled by the NEXT step's launch; a partial reset must not lose them
        KParams Pp = P;
        use_list(h, Pp, (h->parity + 2) % 3);
        launch_mode<MODE_PREFETCH>(h, Pp, nullptr, PREFETCH_MAX_ENVS, s, h->rl_cap[0]);
      }
      HIP_TRY(h, hipMemsetAsync(h->d_rcount, 0, 5 * sizeof(int), s));
      use_list(h, P, 3);
    }
    launch_mode<MODE_RESET>(h, P, nullptr, GROUP, s);
    if (pf) {
      launch_mode<MODE_PREFETCH>(h, P, nullptr, 8, s, h->rl_cap[3]);  // the next two episodes of every env just reset
      if (mask == nullptr) h->dirty_steps = 0;                          // every env now has valid records
    }
    if (mask == nullptr) h->steps_since_full_reset = 0;                 // every step counter is 0: the truncation bursts are now predictable
  } else {
    launch_mode<MODE_REFRESH>(h, P, nullptr, GROUP, s);
  }
  // leave the consumed counter zeroed so the next step can append to either slot
  HIP_TRY(h, hipMemsetAsync(h->buf.done_count, 0, 2 * sizeof(int32_t), s));
  HIP_TRY(h, hipGetLastError());
  return URGYM_OK;
}

}  // namespace

extern "C" {

int urgym_abi_version(void) { return URGYM_ABI_VERSION; }

int urgym_config_default(int env_kind, int num_envs, urgym_config* cfg) {
  if (!cfg || env_kind < 0 || env_kind > 3 || num_envs <= 0) return fail(nullptr, URGYM_ERR_ARG, "urgym_config_default: bad argument");
  fill_default(env_kind, num_envs, cfg);
  return URGYM_OK;
}

int urgym_obs_dims(int env_kind, int* obs_dim, int* goal_dim) {
  if (env_kind < 0 || env_kind > 3 || !obs_dim || !goal_dim) return fail(nullptr, URGYM_ERR_ARG, "urgym_obs_dims: bad argument");
  *obs_dim = env_kind == URGYM_ENV_ORI ? 18 : (env_kind == URGYM_ENV_OBS ? 26 : (env_kind == URGYM_ENV_STA ? 29 : 35));
  *goal_dim = env_kind == URGYM_ENV_OBS ? 3 : 6;
  return URGYM_OK;
}

int urgym_create(const urgym_config* cfg, int device, void** handle) {
  if (!cfg || !handle) return fail(nullptr, URGYM_ERR_ARG, "urgym_create: null argument");
  if (cfg->env_kind < 0 || cfg->env_kind > 3 || cfg->num_envs <= 0) return fail(nullptr, URGYM_ERR_ARG, "urgym_create: bad env_kind/num_envs");
  if (cfg->gjk_start != URGYM_GJK_START_BULLET && cfg->gjk_start != URGYM_GJK_START_GUIDED)
    return fail(nullptr, URGYM_ERR_ARG, "urgym_create: bad gjk_start");
  if (cfg->link_dist_scope != URGYM_LINK_DIST_OBSTACLE && cfg->link_dist_scope != URGYM_LINK_DIST_WORKBENCH)
    return fail(nullptr, URGYM_ERR_ARG, "urgym_create: bad link_dist_scope");
  if (device < 0) return fail(nullptr, URGYM_ERR_ARG, "urgym_create: this library has no CPU path; device must be a HIP ordinal >= 0");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || device >= count) return fail(nullptr, URGYM_ERR_HIP, "urgym_create: HIP device not available", e);
  HIP_TRY(nullptr, hipSetDevice(device));
  Handle* h = new (std::nothrow) Handle();
  if (!h) return fail(nullptr, URGYM_ERR_STATE, "out of memory");
  h->cfg = *cfg;
  h->device = device;
  urgym_obs_dims(cfg->env_kind, &h->obs_dim, &h->goal_dim);
  // constant tables
  DevTables t;
  memcpy(t.joint_rot, UR5E_JOINT_ROT, sizeof(t.joint_rot));
  memcpy(t.joint_xyz, UR5E_JOINT_XYZ, sizeof(t.joint_xyz));
  memcpy(t.capsule, UR5E_CAPSULE, sizeof(t.capsule));
  const HostTables& tabs = build_host_tables();
  if (!tabs.ok) { delete h; return fail(nullptr, URGYM_ERR_STATE, "urgym_create: support map has more records than a 16-bit cell code addresses"); }
  e = hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &t, sizeof(t));
  if (e != hipSuccess) { delete h; return fail(nullptr, URGYM_ERR_HIP, "hipMemcpyToSymbol(c_tab)", e); }
  auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t r = hipMalloc(dst, bytes);
    if (r != hipSuccess) return r;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  e = hipMalloc((void**)&h->d_ld_scratch, sizeof(double) * 5 * (size_t)cfg->num_envs);
  // URGYM_SETUP_CACHE (tuning / tests): 0 = every draw recomputes its operands, 1 (default) = sin / cos of the joints + obstacle pose
  // cached, 2 = the link frames too.  Measured at N = 65536 Dyn (profiles/r2/exp_setup_cache_levels.txt): 186.3 / 195.4 / 197.7 M
  // env-steps/s at 108 / 140 / 201 MB of L2 <-> fabric traffic per launch: the frames buy 1 % for 61 MB, so they stay opt-in.
  const int sc_level = getenv("URGYM_SETUP_CACHE") ? atoi(getenv("URGYM_SETUP_CACHE")) : 1;
  h->sc_frames = sc_level >= 2;
  if (e == hipSuccess && sc_level != 0)
    e = hipMalloc((void**)&h->d_sc_scratch, sizeof(double) * (h->sc_frames ? SC_FIELDS : SC_FRAMES) * (size_t)cfg->num_envs);
  if (e == hipSuccess) e = upload((void**)&h->d_recs, tabs.recs.data(), tabs.recs.size() * sizeof(CandRec));
  if (e == hipSuccess) e = upload((void**)&h->d_cell, tabs.cell.data(), tabs.cell.size() * sizeof(unsigned short));
  if (e != hipSuccess) {
    if (h->d_ld_scratch) hipFree(h->d_ld_scratch);
    if (h->d_sc_scratch) hipFree(h->d_sc_scratch);
    if (h->d_recs) hipFree(h->d_recs);
    if (h->d_cell) hipFree(h->d_cell);
    delete h;
    return fail(nullptr, URGYM_ERR_HIP, "hull table upload", e);
  }
  // Envs per step workgroup (E <= STEP_MAX_ENVS = 128).  Measured on MI355X (DESIGN.md "launch geometry"): the kernel is bound
  // by the latency of the GJK iteration chains; a workgroup's lifetime grows slowly with E (145 us at 46 envs, 177 us at 64),
  // while every additional ROUND of workgroups costs a whole lifetime plus a ragged tail.  So the fewest rounds win:
  //   * N fits one round of <= 128-env workgroups: E = ceil(N / slots), but at least 8, rounded up to a multiple of 8 below
  //     64 (64-byte runs of the float64 state arrays) -- 65536 envs -> 91 per workgroup, all 721 resident at once; 16384 -> 24;
  //   * otherwise R = ceil(N / (128 slots)) rounds of equal workgroups: E = ceil(N / (R slots)).
  // URGYM_STEP_ENVS / URGYM_STEP_TIERS / URGYM_RESET_ENVS override the choices (tuning / tests).
  {
    int cus = 256, per_cu = 3;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    hipError_t oe = hipSuccess;
    // (of the kernel instance the steady-state step really launches: the fused STEP + PREFETCH kernel for the obstacle envs with
    //  auto-reset, the plain STEP kernel otherwise.  The API over-reports near the LDS limit -- DESIGN.md section 4 "toolchain hazards";
    //  the residency census, tools/diag/census.hip, is what the cap URGYM_RESIDENT = 3 below rests on.)
    const bool fused = cfg->env_kind != URGYM_ENV_ORI && cfg->auto_reset && !(getenv("URGYM_PREFETCH") && atoi(getenv("URGYM_PREFETCH")) == 0);
    const bool epa = !cfg->check_collision;
    auto occ = [&](auto kernel) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, THREADS, 0); };
    switch (cfg->env_kind) {
      case URGYM_ENV_ORI: oe = occ(env_kernel<URGYM_ENV_ORI, MODE_STEP, false>); break;
      case URGYM_ENV_OBS: oe = fused ? occ(env_step_fused<URGYM_ENV_OBS, true>) : occ(env_kernel<URGYM_ENV_OBS, MODE_STEP, true>); break;
      case URGYM_ENV_STA:
        if (fused) oe = epa ? occ(env_step_fused<URGYM_ENV_STA, true>) : occ(env_step_fused<URGYM_ENV_STA, false>);
        else oe = epa ? occ(env_kernel<URGYM_ENV_STA, MODE_STEP, true>) : occ(env_kernel<URGYM_ENV_STA, MODE_STEP, false>);
        break;
      default:
        if (fused) oe = epa ? occ(env_step_fused<URGYM_ENV_DYN, true>) : occ(env_step_fused<URGYM_ENV_DYN, false>);
        else oe = epa ? occ(env_kernel<URGYM_ENV_DYN, MODE_STEP, true>) : occ(env_kernel<URGYM_ENV_DYN, MODE_STEP, false>);
        break;
    }
    if (oe != hipSuccess || per_cu < 1) per_cu = 3;
    if (per_cu > URGYM_RESIDENT) per_cu = URGYM_RESIDENT;
    long slots = (long)cus * per_cu;
    const long n = cfg->num_envs;
    // prefetched episode records (below): the refill of ~2 % of the envs runs beside the step kernel, 32 envs per workgroup
    bool want_prefetch = cfg->env_kind != URGYM_ENV_ORI;
    if (const char* ov = getenv("URGYM_PREFETCH")) want_prefetch = want_prefetch && atoi(ov) != 0;  // (Ori: inline reset, below)
    if (want_prefetch && cfg->auto_reset) {
      const long refill = n / 1600;
      slots -= refill < slots / 8 ? refill : slots / 8;
    }
    const long rounds = (n + slots * STEP_MAX_ENVS - 1) / (slots * STEP_MAX_ENVS);
    long envs = (n + slots * rounds - 1) / (slots * rounds);
    if (envs < 8) envs = 8;
    if (envs < GROUP) envs = (envs + 7) / 8 * 8;
    if (envs > STEP_MAX_ENVS) envs = STEP_MAX_ENVS;
    h->step_envs = (int)envs;
    // One round, three workgroups per CU: the first 2 x CUs workgroups (two per CU, dispatched first) serve E1 envs each, the third
    // one of a CU 0.7 E1.  A CU with three resident workgroups advances each of them more slowly than one with two, and the third
    // starts last; giving it less work evens the finishing times out (N = 65536: 512 x 100 + 205 x 70 instead of 721 x 91, +5 %,
    // profiles/r2/exp_two_tier_one_round.jsonl).  URGYM_STEP_TIERS=0 keeps the uniform geometry.
    if (rounds == 1 && per_cu == 3 && slots > 2L * cus) {
      const long big = 2L * cus, rest = slots - big;
      const long e1 = (10 * n + (10 * big + 7 * rest) - 1) / (10 * big + 7 * rest);
      if (e1 >= 64 && e1 <= STEP_MAX_ENVS && n > big * e1) {  // (below ~40 000 envs the uniform geometry is as fast or faster)
        const long e2 = (n - big * e1 + rest - 1) / rest;
        h->step_envs = (int)e1; h->big_blocks = (int)big; h->tail_envs = (int)(e2 < 1 ? 1 : e2);
      }
    }
    // auto-reset kernel: ~1 % of the envs finish per step; keep that to about one workgroup per CU (4 envs at N = 65536,
    // 8 at 262144): it is pure latency, smaller workgroups shorten the wave-wide maxima, more than one per CU queue up
    int renvs = 4;
    while (renvs < GROUP && n / 100 > (long)renvs * (cus + cus / 4)) renvs *= 2;
    h->reset_envs = renvs;
    if (const char* ov = getenv("URGYM_STEP_ENVS")) {
      const int v = atoi(ov);
      if (v >= 1 && v <= STEP_MAX_ENVS) { h->step_envs = v; h->big_blocks = 0; h->tail_envs = 0; }
    }
    if (const char* ov = getenv("URGYM_STEP_TIERS")) {  // "E1,B,E2": B workgroups of E1 envs, then workgroups of E2 (tuning / tests)
      int e1 = 0, b = 0, e2 = 0;
      if (sscanf(ov, "%d,%d,%d", &e1, &b, &e2) == 3 && e1 >= 1 && e1 <= STEP_MAX_ENVS && e2 >= 1 && e2 <= STEP_MAX_ENVS && b >= 1) {
        h->step_envs = e1; h->big_blocks = b; h->tail_envs = e2;
      } else if (atoi(ov) == 0) {  // "0": uniform workgroups
        h->step_envs = (int)envs; h->big_blocks = 0; h->tail_envs = 0;
      }
    }
    if (const char* ov = getenv("URGYM_REFILL_BLOCKS")) {
      const int r = atoi(ov);
      if (r >= 1) h->refill_blocks_override = r;
    }
    if (const char* ov = getenv("URGYM_RESET_ENVS")) {
      const int r = atoi(ov);
      if (r >= 1 && r <= MAX_ENVS) h->reset_envs = r;
    }
    // prefetched episode records: on unless URGYM_PREFETCH=0 (then finished envs are reset by a kernel after each step)
    // (Ori's reset is a goal draw, no distance query: there the extra launches cost more than the reset kernel they replace)
    {  // the neutral pose's end-effector frame (the first six slots of every reset observation), by the device code itself
      double* dq = nullptr;
      float* dout = nullptr;
      hipError_t ne = hipMalloc((void**)&dq, sizeof(double) * 6);
      if (ne == hipSuccess) ne = hipMalloc((void**)&dout, sizeof(float) * 6);
      if (ne == hipSuccess) ne = hipMemcpy(dq, cfg->neutral_q, sizeof(double) * 6, hipMemcpyHostToDevice);
      if (ne == hipSuccess) {
        hipLaunchKernelGGL(ee_pose_kernel, dim3(1), dim3(1), 0, 0, dq, dout);
        ne = hipMemcpy(h->neutral_ach, dout, sizeof(float) * 6, hipMemcpyDeviceToHost);
      }
      if (dq) hipFree(dq);
      if (dout) hipFree(dout);
      // UR5OriReach-v1: inline reset unless URGYM_PREFETCH=0 asks for the reset kernel (the switch of the obstacle envs, same meaning)
      bool want_inline = cfg->env_kind == URGYM_ENV_ORI && ne == hipSuccess;
      if (const char* ov = getenv("URGYM_PREFETCH")) want_inline = want_inline && atoi(ov) != 0;
      h->inline_ori = want_inline;
    }
    h->prefetch = want_prefetch;
    if (h->prefetch) {
      const size_t nn = (size_t)n;
      h->rl_cap[0] = h->rl_cap[1] = h->rl_cap[2] = (int)n;   // at most one entry per env and step: no entry is ever dropped
      h->rl_cap[3] = (int)(2 * n);
      h->dirty_steps = cfg->max_episode_steps + 1;             // no record exists yet
      hipError_t pe = hipMalloc((void**)&h->d_rec, sizeof(double) * 2 * REC_FIELDS * nn);
      if (pe == hipSuccess) pe = hipMalloc((void**)&h->d_reci, sizeof(int32_t) * 4 * nn);
      for (int i = 0; i < 4 && pe == hipSuccess; i++) pe = hipMalloc((void**)&h->d_rl[i], sizeof(int2) * (size_t)h->rl_cap[i]);
      if (pe == hipSuccess) pe = hipMalloc((void**)&h->d_rcount, sizeof(int) * 5);
      if (pe == hipSuccess) pe = hipMemset(h->d_reci, 0xFF, sizeof(int32_t) * 4 * nn);
      if (pe == hipSuccess) pe = hipMemset(h->d_rcount, 0, sizeof(int) * 5);
      if (pe != hipSuccess) {
        release_prefetch(h);
        if (h->d_ld_scratch) hipFree(h->d_ld_scratch);
        if (h->d_sc_scratch) hipFree(h->d_sc_scratch);
        if (h->d_recs) hipFree(h->d_recs);
        if (h->d_cell) hipFree(h->d_cell);
        delete h;
        return fail(nullptr, URGYM_ERR_HIP, "prefetch buffers", pe);
      }
    }
    if (getenv("URGYM_VERBOSE"))
      fprintf(stderr, "[urgym] device %d: %d CUs x %d resident step workgroups; N = %ld -> %d envs per step workgroup (%d of them, then %d envs each), %d per reset workgroup\n",
              device, cus, per_cu, n, h->step_envs, h->big_blocks > 0 ? h->big_blocks : (int)((n + h->step_envs - 1) / h->step_envs), h->tail_envs, h->reset_envs);
    if (getenv("URGYM_VERBOSE")) fprintf(stderr, "[urgym] prefetched episode records: %s\n", h->prefetch ? "on" : "off");
  }
  *handle = h;
  return URGYM_OK;
}

int urgym_destroy(void* handle) {
  Handle* h = (Handle*)handle;
  if (!h) return URGYM_OK;
  hipSetDevice(h->device);
  release_prefetch(h);
  for (auto e : h->ev) hipEventDestroy(e);
  if (h->d_ld_scratch) hipFree(h->d_ld_scratch);
  if (h->d_sc_scratch) hipFree(h->d_sc_scratch);
  if (h->d_recs) hipFree(h->d_recs);
  if (h->d_cell) hipFree(h->d_cell);
  delete h;
  return URGYM_OK;
}

int urgym_bind(void* handle, const urgym_buffers* b) {
  Handle* h = (Handle*)handle;
  if (!h || !b) return fail(h, URGYM_ERR_ARG, "urgym_bind: null argument");
  const bool obst = h->cfg.env_kind != URGYM_ENV_ORI;
  if (!b->q || !b->goal || !b->step_count || !b->episode_id || !b->observation || !b->achieved_goal || !b->desired_goal ||
      !b->reward || !b->terminated || !b->truncated || !b->is_success || !b->collision || !b->final_observation ||
      !b->final_achieved_goal || !b->final_desired_goal || !b->status || !b->done_list || !b->done_count)
    return fail(h, URGYM_ERR_ARG, "urgym_bind: a required buffer pointer is null");
  if (obst && (!b->obst_start || !b->obst_end || !b->obst_pos || !b->obst_quat || !b->obst_vel || !b->link_dist))
    return fail(h, URGYM_ERR_ARG, "urgym_bind: an obstacle buffer pointer is null");
  if (h->prefetch) {  // records belong to the state that was bound before
    hipSetDevice(h->device);
    HIP_TRY(h, hipMemset(h->d_reci, 0xFF, sizeof(int32_t) * 4 * (size_t)h->cfg.num_envs));
    HIP_TRY(h, hipMemset(h->d_rcount, 0, sizeof(int) * 5));
    h->rec_seed_valid = false;
    h->dirty_steps = h->cfg.max_episode_steps + 1;
  }
  h->steps_since_full_reset = -1;
  h->buf = *b;
  h->bound = true;
  return URGYM_OK;
}

int urgym_reset(void* handle, const uint8_t* mask_dev, uint64_t seed, void* stream) {
  Handle* h = (Handle*)handle;
  int rc = check_bound(h);
  if (rc) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  if (seed != UINT64_MAX) h->seed = seed;
  return do_masked(h, mask_dev, MODE_RESET, (hipStream_t)stream);
}

int urgym_invalidate_records(void* handle) {
  Handle* h = (Handle*)handle;
  if (!h) return fail(nullptr, URGYM_ERR_ARG, "null handle");
  if (h->prefetch) {
    // Really invalidate (like urgym_bind): wipe every record key and every pending refill entry, so that EVERY env takes the
    // kernel fallback at its first finish and comes out of it with fresh records for its next two episodes -- which is what makes the
    // max_episode_steps + 1 window sufficient.  Merely opening the window is not: after an episode_id edit one of an env's two slots
    // can still carry a key that matches again later (the env would then be reset inline from it once the window has closed while
    // its other slot is stale), and a refill entry filed before the edit could rewrite a slot the step is reading.
    // The caller has just edited the bound buffers from the host side, i.e. between launches: wait for whatever is in flight.
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemset(h->d_reci, 0xFF, sizeof(int32_t) * 4 * (size_t)h->cfg.num_envs));
    HIP_TRY(h, hipMemset(h->d_rcount, 0, sizeof(int) * 5));
  }
  h->dirty_steps = h->cfg.max_episode_steps + 1;
  h->steps_since_full_reset = -1;  // step counters were edited: no longer known when many envs finish at once
  return URGYM_OK;
}

int urgym_derive_obstacle_motion(void* handle, void* stream) {
  Handle* h = (Handle*)handle;
  int rc = check_bound(h);
  if (rc) return rc;
  if (h->cfg.env_kind == URGYM_ENV_ORI) return URGYM_OK;  // no obstacle
  HIP_TRY(h, hipSetDevice(h->device));
  const int N = h->cfg.num_envs;
  hipLaunchKernelGGL(derive_displacement_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->buf.obst_vel, N, h->cfg.dt);
  HIP_TRY(h, hipGetLastError());
  return URGYM_OK;
}

int urgym_refresh(void* handle, const uint8_t* mask_dev, void* stream) {
  Handle* h = (Handle*)handle;
  int rc = check_bound(h);
  if (rc) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  return do_masked(h, mask_dev, MODE_REFRESH, (hipStream_t)stream);
}

int urgym_step(void* handle, const float* actions_dev, void* stream) {
  Handle* h = (Handle*)handle;
  int rc = check_bound(h);
  if (rc) return rc;
  if (!actions_dev) return fail(h, URGYM_ERR_ARG, "urgym_step: null actions");
  HIP_TRY(h, hipSetDevice(h->device));
  return do_step(h, actions_dev, (hipStream_t)stream);
}

int urgym_rollout(void* handle, const float* actions_dev, int num_steps, void* stream) {
  Handle* h = (Handle*)handle;
  int rc = check_bound(h);
  if (rc) return rc;
  if (!actions_dev || num_steps < 0) return fail(h, URGYM_ERR_ARG, "urgym_rollout: bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  for (int k = 0; k < num_steps; k++) {
    rc = do_step(h, actions_dev + (size_t)k * h->cfg.num_envs * 6, (hipStream_t)stream);
    if (rc) return rc;
  }
  return URGYM_OK;
}

int urgym_probe_closest(void* handle, int count, const int* type_a, const double* par_a, const double* pose_a, const int* type_b,
                        const double* par_b, const double* pose_b, double threshold, double* out_dist, int* out_info, void* stream) {
  Handle* h = (Handle*)handle;
  if (!h || count < 0) return fail(h, URGYM_ERR_ARG, "urgym_probe_closest: bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  if (count == 0) return URGYM_OK;
  HullMap g;
  g.recs = h->d_recs; g.cell = h->d_cell;
  hipLaunchKernelGGL(probe_closest_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, g, count, type_a, par_a, pose_a,
                     type_b, par_b, pose_b, threshold, out_dist, out_info);
  HIP_TRY(h, hipGetLastError());
  return URGYM_OK;
}

int urgym_probe_pose_distance(void* handle, int count, const double* a6, const double* b6, double* out2, void* stream) {
  Handle* h = (Handle*)handle;
  if (!h || count < 0 || (count > 0 && (!a6 || !b6 || !out2))) return fail(h, URGYM_ERR_ARG, "urgym_probe_pose_distance: bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  if (count == 0) return URGYM_OK;
  hipLaunchKernelGGL(probe_pose_distance_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, count, a6, b6, out2);
  HIP_TRY(h, hipGetLastError());
  return URGYM_OK;
}

int urgym_enable_timing(void* handle, int enable) {
  Handle* h = (Handle*)handle;
  if (!h) return fail(nullptr, URGYM_ERR_ARG, "null handle");
  h->timing = enable != 0;
  h->timing_every = enable > 1 ? enable : 1;
  h->timing_tick = 0;
  h->ev_used = 0;
  return URGYM_OK;
}

int urgym_query_timing(void* handle, double* step_us, double* reset_us, int* launches) {
  Handle* h = (Handle*)handle;
  if (!h) return fail(nullptr, URGYM_ERR_ARG, "null handle");
  if (!h->timing) return fail(h, URGYM_ERR_STATE, "timing not enabled");
  double acc[3] = {0, 0, 0};
  int cnt[3] = {0, 0, 0};
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    HIP_TRY(h, hipEventSynchronize(h->ev[i + 1]));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    int k = h->ev_kind[i / 2];
    acc[k] += (double)ms * 1000.0;
    cnt[k]++;
  }
  if (step_us) *step_us = cnt[0] ? acc[0] / cnt[0] : 0.0;
  if (reset_us) *reset_us = cnt[1] ? acc[1] / cnt[1] : 0.0;
  h->last_refill_us = cnt[2] ? acc[2] / cnt[2] : 0.0;
  if (launches) *launches = cnt[0];
  h->ev_used = 0;
  return URGYM_OK;
}

#ifdef URGYM_STAMPS
int urgym_debug_occupancy(int* blocks_per_cu) {
  return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, env_kernel<URGYM_ENV_DYN, MODE_STEP, false>, THREADS, 0);
}
int urgym_debug_stamps(unsigned long long* out, int count) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)count, 0, hipMemcpyDeviceToHost);
}
#endif

int urgym_query_refill_timing(void* handle, double* refill_us) {
  Handle* h = (Handle*)handle;
  if (!h || !refill_us) return fail(h, URGYM_ERR_ARG, "urgym_query_refill_timing: null argument");
  *refill_us = h->last_refill_us;
  return URGYM_OK;
}

const char* urgym_last_error(void* handle) {
  Handle* h = (Handle*)handle;
  return h ? h->err : g_err;
}

}  // extern "C"
