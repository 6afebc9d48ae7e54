// urgym_tables_host.h — host-side construction of the device lookup tables from data/ur5e_model.h
// (included by urgym_hip.hip and by the CPU test harness tests/device_harness.cpp).
#pragma once
#include <algorithm>
#include <vector>

#include "../../data/ur5e_model.h"
#include "urgym_device.h"

namespace urgym {

struct HostTables {
  std::vector<NbrRec> recs;
  std::vector<unsigned short> dirmap;  // [6][DIRMAP_CELLS]
};

inline HostTables make_host_tables() {
  HostTables t;
  const int NV = UR5E_NUM_HULL_VERTS;
  t.recs.resize(NV);
  auto set_slot = [&](NbrRec& r, int j, int v) {
    r.id.v[j] = (unsigned short)v;
    double* xs = &r.x[0].a; double* ys = &r.y[0].a; double* zs = &r.z[0].a;
    xs[j] = UR5E_HULL_VERTS[v][0]; ys[j] = UR5E_HULL_VERTS[v][1]; zs[j] = UR5E_HULL_VERTS[v][2];
  };
  // record chain of vertex i: the vertex ITSELF and its neighbours, sorted by DESCENDING id, eight per record; the unused slots
  // of the last record repeat the lowest id.  The device examines the entries in this order and keeps a candidate when its value
  // is >= the best so far, so among exactly tied values the LOWEST id wins -- the vertex the oracle's scan (first maximum) returns.
  for (int i = 0; i < NV; i++) {
    std::vector<int> ent(UR5E_ADJ_INDEX + UR5E_ADJ_OFFSET[i], UR5E_ADJ_INDEX + UR5E_ADJ_OFFSET[i + 1]);
    ent.push_back(i);
    std::sort(ent.begin(), ent.end(), [](int x, int y) { return x > y; });
    int rec = i;
    for (size_t base = 0; base < ent.size(); base += 8) {
      if (base > 0) {  // chain an overflow record
        t.recs.push_back(NbrRec{});
        t.recs[rec].next = (int)t.recs.size() - 1;
        rec = (int)t.recs.size() - 1;
      }
      t.recs[rec].next = -1;
      t.recs[rec].pad[0] = t.recs[rec].pad[1] = t.recs[rec].pad[2] = 0;
      for (int j = 0; j < 8; j++) set_slot(t.recs[rec], j, ent[std::min(base + j, ent.size() - 1)]);
    }
  }
  // direction map: the support vertex of every cell-centre direction.  The map only picks where the device's exact climb starts,
  // so each cell is filled by the same kind of climb on the adjacency lists, started from the neighbouring cell's answer (a brute-
  // force scan per cell would cost seconds at urgym_create for the 6 x 6 x G x G cells).
  const int G = DIRMAP_G;
  t.dirmap.resize((size_t)6 * DIRMAP_CELLS);
  auto value = [](int k, const double* d) {
    return (UR5E_HULL_VERTS[k][0] * d[0] + UR5E_HULL_VERTS[k][1] * d[1]) + UR5E_HULL_VERTS[k][2] * d[2];
  };
  for (int h = 0; h < 6; h++)
    for (int face = 0; face < 6; face++) {
      int row_start = -1;
      for (int iv = 0; iv < G; iv++) {
        int cur = row_start;
        for (int iu = 0; iu < G; iu++) {
          const int axis = face / 2;
          double d[3];
          d[axis] = (face & 1) ? -1.0 : 1.0;
          d[(axis + 1) % 3] = (iu + 0.5) / G * 2.0 - 1.0;
          d[(axis + 2) % 3] = (iv + 0.5) / G * 2.0 - 1.0;
          if (cur < 0) {  // first cell of the face: scan
            cur = UR5E_HULL_OFFSET[h];
            for (int k = UR5E_HULL_OFFSET[h]; k < UR5E_HULL_OFFSET[h + 1]; k++)
              if (value(k, d) > value(cur, d)) cur = k;
          }
          for (;;) {  // steepest ascent over the neighbours until none is better
            int nxt = cur;
            double bv = value(cur, d);
            for (int e = UR5E_ADJ_OFFSET[cur]; e < UR5E_ADJ_OFFSET[cur + 1]; e++) {
              const double val = value(UR5E_ADJ_INDEX[e], d);
              if (val > bv) { bv = val; nxt = UR5E_ADJ_INDEX[e]; }
            }
            if (nxt == cur) break;
            cur = nxt;
          }
          if (iu == 0) row_start = cur;
          t.dirmap[(size_t)h * DIRMAP_CELLS + (face * G + iv) * G + iu] = (unsigned short)cur;
        }
      }
    }
  return t;
}

// built once per process (a few milliseconds; every handle uploads its own device copy)
inline const HostTables& build_host_tables() {
  static const HostTables tabs = make_host_tables();
  return tabs;
}

}  // namespace urgym
