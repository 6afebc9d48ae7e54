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
  auto fill = [&](NbrRec& r, int self) {
    r.next = -1;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    for (int j = 0; j < 8; j++) {
      r.id[j] = (unsigned short)self;
      r.x[j] = UR5E_HULL_VERTS[self][0]; r.y[j] = UR5E_HULL_VERTS[self][1]; r.z[j] = UR5E_HULL_VERTS[self][2];
    }
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
      NbrRec& r = t.recs[rec];
      r.id[slot] = (unsigned short)nb;
      r.x[slot] = UR5E_HULL_VERTS[nb][0]; r.y[slot] = UR5E_HULL_VERTS[nb][1]; r.z[slot] = UR5E_HULL_VERTS[nb][2];
      slot++;
    }
  }
  for (int h = 0; h < 6; h++)
    for (int s = 0; s < HULL_SEEDS; s++) {
      const int v = UR5E_SEEDS[h][s];
      t.seeds[h].id[s] = v;
      t.seeds[h].x[s] = UR5E_HULL_VERTS[v][0]; t.seeds[h].y[s] = UR5E_HULL_VERTS[v][1]; t.seeds[h].z[s] = UR5E_HULL_VERTS[v][2];
    }
  return t;
}

}  // namespace urgym
