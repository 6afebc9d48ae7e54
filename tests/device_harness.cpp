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
  HullMap g{tabs.recs.data(), tabs.cell.data()};
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

// Census of the exact support map against the linear scan the oracle runs (first maximum of (x dx + y dy) + z dz over the hull's
// vertices in id order): `count` directions per hull.  mode 0: random directions; 1: directions in which two NEIGHBOURING vertices
// tie (boundaries of the normal cones, up to rounding); 2: exact face normals of the hull's triangles (three or more cones meet;
// coplanar faces: many); 3: mode 2 perturbed by 1e-9; 4: axis-aligned and cube-map edge / corner directions (cell borders, face
// switches of the cube map).  out[0] = mismatching coordinates, out[1] = directions tested, out[2] = records visited in total.
extern "C" int harness_support_census(int mode, int count, unsigned long long seed, long* out) {
  const HostTables& tabs = build_host_tables();
  if (!tabs.ok) return -1;
  HullMap g{tabs.recs.data(), tabs.cell.data()};
  unsigned long long st = seed * 6364136223846793005ULL + 1442695040888963407ULL;
  auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
  auto gauss = [&]() { double a = 0; for (int i = 0; i < 12; i++) a += rnd(); return a - 6.0; };
  long bad = 0, tested = 0, visited = 0;
  for (int h = 0; h < 6; h++) {
    const int v0 = UR5E_HULL_OFFSET[h], v1 = UR5E_HULL_OFFSET[h + 1];
    for (int it = 0; it < count; it++) {
      double d[3] = {gauss(), gauss(), gauss()};
      const int p = v0 + (int)(rnd() * (v1 - v0));
      const int ne = UR5E_ADJ_OFFSET[p + 1] - UR5E_ADJ_OFFSET[p];
      const int n1 = UR5E_ADJ_INDEX[UR5E_ADJ_OFFSET[p] + (int)(rnd() * ne)];
      auto sub = [&](int a, int b, double* e) { for (int k = 0; k < 3; k++) e[k] = UR5E_HULL_VERTS[a][k] - UR5E_HULL_VERTS[b][k]; };
      if (mode == 1) {
        double e[3];
        sub(p, n1, e);
        const double k = (d[0] * e[0] + d[1] * e[1] + d[2] * e[2]) / (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        for (int i = 0; i < 3; i++) d[i] -= k * e[i];
      } else if (mode == 2 || mode == 3) {
        // a triangle of the surface graph: p, n1 and a common neighbour n2; its outward normal
        int n2 = -1;
        for (int a = UR5E_ADJ_OFFSET[p]; a < UR5E_ADJ_OFFSET[p + 1] && n2 < 0; a++)
          for (int b = UR5E_ADJ_OFFSET[n1]; b < UR5E_ADJ_OFFSET[n1 + 1]; b++)
            if (UR5E_ADJ_INDEX[a] == UR5E_ADJ_INDEX[b]) { n2 = UR5E_ADJ_INDEX[a]; break; }
        if (n2 < 0) continue;
        double e1[3], e2[3];
        sub(n1, p, e1); sub(n2, p, e2);
        double nrm[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        // orient outward: away from the hull's centroid-ish point (the mean of a few vertices is inside)
        double c[3] = {0, 0, 0};
        for (int k = v0; k < v1; k++) for (int i = 0; i < 3; i++) c[i] += UR5E_HULL_VERTS[k][i] / (v1 - v0);
        const double side = nrm[0] * (UR5E_HULL_VERTS[p][0] - c[0]) + nrm[1] * (UR5E_HULL_VERTS[p][1] - c[1]) + nrm[2] * (UR5E_HULL_VERTS[p][2] - c[2]);
        const double len = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        if (!(len > 0.0)) continue;
        for (int i = 0; i < 3; i++) d[i] = (side < 0 ? -nrm[i] : nrm[i]) / len + (mode == 3 ? 1e-9 * d[i] : 0.0);
      } else if (mode == 4) {
        const int G = DIRMAP_G;
        const int kind = (int)(rnd() * 4);
        const double s0 = rnd() < 0.5 ? -1.0 : 1.0, s1 = rnd() < 0.5 ? -1.0 : 1.0;
        const int ax = (int)(rnd() * 3);
        double u = (kind == 0) ? 0.0 : (kind == 1 ? s1 : (2.0 * (int)(rnd() * (G + 1)) / G - 1.0));  // axis / cube edge / cell border
        double v = (kind == 3) ? (2.0 * (int)(rnd() * (G + 1)) / G - 1.0) : (kind == 1 ? (rnd() < 0.3 ? s0 : 2.0 * rnd() - 1.0) : (kind == 0 ? 0.0 : 2.0 * rnd() - 1.0));
        const double scale = std::exp(6.0 * rnd() - 3.0);
        d[ax] = s0 * scale; d[(ax + 1) % 3] = u * scale; d[(ax + 2) % 3] = v * scale;
      }
      int best = v0;
      double bv = -1.0e300;
      for (int k = v0; k < v1; k++) {
        const double x = (UR5E_HULL_VERTS[k][0] * d[0] + UR5E_HULL_VERTS[k][1] * d[1]) + UR5E_HULL_VERTS[k][2] * d[2];
        if (x > bv) { bv = x; best = k; }
      }
      const D3 got = hull_support(g, h, d3(d[0], d[1], d[2]));
      for (int rec = tabs.cell[(size_t)h * DIRMAP_CELLS + dirmap_cell(d3(d[0], d[1], d[2]))]; rec >= 0; rec = tabs.recs[rec].next) visited++;
      tested++;
      if (got.x != UR5E_HULL_VERTS[best][0] || got.y != UR5E_HULL_VERTS[best][1] || got.z != UR5E_HULL_VERTS[best][2]) bad++;
    }
  }
  out[0] = bad; out[1] = tested; out[2] = visited;
  return 0;
}

// statistics of the table build: out[0..5] = cells with 1, 2, 3, 4, 5..8, > 8 candidates; out[6] = records; out[7] = longest list
extern "C" int harness_table_stats(long* out) {
  const HostTables& tabs = build_host_tables();
  for (int i = 0; i < 6; i++) out[i] = tabs.cells_by_candidates[i];
  out[6] = (long)tabs.recs.size();
  out[7] = tabs.longest_list;
  return tabs.ok ? 0 : -1;
}
