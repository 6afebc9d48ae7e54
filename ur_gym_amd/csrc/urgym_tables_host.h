// urgym_tables_host.h — host-side construction of the device lookup tables from data/ur5e_model.h
// (included by urgym_hip.hip and by the CPU test harness tests/device_harness.cpp).
#pragma once
#include <vector>

#include "../../data/ur5e_model.h"
#include "urgym_device.h"

namespace urgym {

struct HostTables {
  std::vector<NbrRec> recs;
  std::vector<unsigned short> dirmap;  // [6][DIRMAP_CELLS]
};

inline HostTables build_host_tables() {
  HostTables t;
  const int NV = UR5E_NUM_HULL_VERTS;
  t.recs.resize(NV);
  auto set_slot = [&](NbrRec& r, int j, int v) {
    r.id.v[j] = (unsigned short)v;
    double* xs = &r.x[0].a; double* ys = &r.y[0].a; double* zs = &r.z[0].a;
    xs[j] = UR5E_HULL_VERTS[v][0]; ys[j] = UR5E_HULL_VERTS[v][1]; zs[j] = UR5E_HULL_VERTS[v][2];
  };
  auto fill = [&](NbrRec& r, int self) {
    r.next = -1;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    for (int j = 0; j < 8; j++) set_slot(r, j, self);
  };
  for (int i = 0; i < NV; i++) fill(t.recs[i], i);
  for (int i = 0; i < NV; i++) {
    int rec = i, slot = 0;
    for (int e = UR5E_ADJ_OFFSET[i]; e < UR5E_ADJ_OFFSET[i + 1]; e++) {
      if (slot == 8) {  // chain an overflow record
        NbrRec extra;
        fill(extra, i);
        t.recs.push_back(extra);
        t.recs[rec].next = (int)t.recs.size() - 1;
        rec = (int)t.recs.size() - 1;
        slot = 0;
      }
      const int nb = UR5E_ADJ_INDEX[e];
      set_slot(t.recs[rec], slot, nb);
      slot++;
    }
  }
  // direction map: support vertex (exact float64 scan, first maximum like Bullet's) of every cell-centre direction
  const int G = DIRMAP_G;
  t.dirmap.resize((size_t)6 * DIRMAP_CELLS);
  for (int h = 0; h < 6; h++)
    for (int face = 0; face < 6; face++)
      for (int iv = 0; iv < G; iv++)
        for (int iu = 0; iu < G; iu++) {
          const int axis = face / 2;
          double d[3];
          d[axis] = (face & 1) ? -1.0 : 1.0;
          d[(axis + 1) % 3] = (iu + 0.5) / G * 2.0 - 1.0;
          d[(axis + 2) % 3] = (iv + 0.5) / G * 2.0 - 1.0;
          int best = UR5E_HULL_OFFSET[h];
          double bv = -1.0e300;
          for (int k = UR5E_HULL_OFFSET[h]; k < UR5E_HULL_OFFSET[h + 1]; k++) {
            const double val = (UR5E_HULL_VERTS[k][0] * d[0] + UR5E_HULL_VERTS[k][1] * d[1]) + UR5E_HULL_VERTS[k][2] * d[2];
            if (val > bv) { bv = val; best = k; }
          }
          t.dirmap[(size_t)h * DIRMAP_CELLS + (face * G + iv) * G + iu] = (unsigned short)best;
        }
  return t;
}

}  // namespace urgym
