// Host harness: the DEVICE GJK (ur_gym_amd/csrc/urgym_device.h) compiled with g++ so that tests can run it on the CPU
// against the oracle over hundreds of thousands of random queries.  Test infrastructure only.
#define URGYM_HOST_HARNESS 1
#include "../ur_gym_amd/csrc/urgym_device.h"
#include "../ur_gym_amd/csrc/urgym_tables_host.h"

using namespace urgym;

static X3 pose_to_x3(const double* p) {
  X3 T;
  quat_to_rot(Q4{p[3], p[4], p[5], p[6]}, T.r);
  T.t = d3(p[0], p[1], p[2]);
  return T;
}
static ShapeDesc desc(int type, const double* par, double* margin) {
  ShapeDesc s;
  s.type = type; s.hull = 0; s.hx = s.hy = s.hz = 0;
  const double m = 0.001;  // Bullet's default collision margin on every createCollisionShape primitive (urgym_hip.hip M_PRIM)
  if (type == SH_HULL) { s.hull = (int)par[0] - 1; *margin = 0.001; }
  else if (type == SH_CYLZ) { s.hx = s.hy = par[0] - m; s.hz = 0.5 * par[1] - m; *margin = m; }
  else if (type == SH_BOX) { s.hx = par[0] - m; s.hy = par[1] - m; s.hz = par[2] - m; *margin = m; }
  else { *margin = par[0]; }
  return s;
}

extern "C" int harness_closest(int type_a, const double* par_a, const double* pose_a, int type_b, const double* par_b,
                               const double* pose_b, double threshold, double* out) {
  const HostTables& tabs = build_host_tables();
  HullGraph g{&UR5E_HULL_VERTS[0][0], tabs.recs.data(), tabs.dirmap.data()};
  double ma, mb;
  ShapeDesc A = desc(type_a, par_a, &ma), B = desc(type_b, par_b, &mb);
  X3 Ta = pose_to_x3(pose_a), Tb = pose_to_x3(pose_b);
  int info;
  double slot[GJK_SLOT_DOUBLES];
  XRef Tr{slot, 1};
  store(Tr, rel(Tb, Ta));
  double core = gjk_core_distance(g, A, Tr, B, rotT(Tb, d3(0, 1, 0)), ma + mb + 0.02 + threshold, info);
  out[0] = core - ma - mb;
  out[1] = info;
  return 0;
}
