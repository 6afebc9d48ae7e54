// urgym_tables_host.h — host-side construction of the device lookup tables from data/ur5e_model.h
// (included by urgym_hip.hip and by the CPU test harness tests/device_harness.cpp).
#pragma once
#include <vector>

#include "../../data/ur5e_model.h"
#include "urgym_device.h"

namespace urgym {

struct HostTables {
  std::vector<NbrRec> recs;
  SeedRec seeds[6];
};

inline HostTables build_host_tables() {
  static_assert(UR5E_NUM_SEEDS == HULL_SEEDS, "seed table width");
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
  for (int h = 0; h < 6; h++)
    for (int s = 0; s < HULL_SEEDS; s++) {
      const int v = UR5E_SEEDS[h][s];
      SeedRec& S = t.seeds[h];
      S.id[s / 8].v[s % 8] = (unsigned short)v;
      (&S.x[0].a)[s] = UR5E_HULL_VERTS[v][0]; (&S.y[0].a)[s] = UR5E_HULL_VERTS[v][1]; (&S.z[0].a)[s] = UR5E_HULL_VERTS[v][2];
    }
  return t;
}

}  // namespace urgym
