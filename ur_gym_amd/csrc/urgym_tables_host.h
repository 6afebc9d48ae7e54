// urgym_tables_host.h — host-side construction of the device lookup tables from data/ur5e_model.h
// (included by urgym_hip.hip and by the CPU test harness tests/device_harness.cpp).
#pragma once
#include <algorithm>
#include <map>
#include <thread>
#include <vector>

#include "../../data/ur5e_model.h"
#include "urgym_device.h"

namespace urgym {

struct HostTables {
  std::vector<CandRec> recs;           // candidate records (urgym_device.h "exact support map")
  std::vector<unsigned short> cell;    // [6 hulls][DIRMAP_CELLS] -> first record of the cell's candidate list
  bool ok = true;                      // false: more records than a 16-bit cell code can address (cannot happen with the shipped hulls)
  // statistics of the build (tests, URGYM_VERBOSE)
  long cells_by_candidates[6] = {0, 0, 0, 0, 0, 0};  // cells with 1, 2, 3, 4, 5..8, > 8 candidates
  int longest_list = 0;
};

// Candidate vertices of one direction cell: every vertex of hull h whose normal cone meets the cell.
//   cell      = the square [u0, u1] x [v0, v1] of cube-map face (axis, s): directions d with d[axis] = s, d[axis+1] = u, d[axis+2] = v
//               (the gnomonic plane of the face: great circles are straight lines there), inflated by `margin` on every side --
//               the device picks the cell with float32 arithmetic;
//   cone of p = { d : d . (p - n) >= -eps for every neighbour n of p in the hull's surface graph }: on a convex polytope a vertex that
//               beats its neighbours beats every vertex, so these are the directions p is the support vertex for; eps > 0 keeps
//               every vertex that is within rounding of the maximum (exact ties on the cone boundaries, coplanar faces).
// The cones that meet a connected region are connected in the graph, so a breadth-first search from the support vertex of the
// cell's centre finds them all; a cone meets the square iff clipping the square by the cone's half-planes leaves something.
// tests/test_device_header_on_host.py checks the result against the linear scan for millions of directions, incl. exact ties.
struct CellCandidates {
  static constexpr double MARGIN = 1.0e-5, EPS = 1.0e-11;
  struct P2 { double u, v; };
  // clip the convex polygon poly[0..n) by a u + b v + c >= -EPS; returns the new vertex count (<= n + 1)
  static int clip(P2* poly, int n, double a, double b, double c) {
    P2 out[40];
    int m = 0;
    for (int i = 0; i < n; i++) {
      const P2 p = poly[i], q = poly[i + 1 == n ? 0 : i + 1];
      const double fp = a * p.u + b * p.v + c + EPS, fq = a * q.u + b * q.v + c + EPS;
      if (fp >= 0.0 && m < 40) out[m++] = p;
      if ((fp >= 0.0) != (fq >= 0.0) && m < 40) {
        const double t = fp / (fp - fq);
        out[m++] = P2{p.u + t * (q.u - p.u), p.v + t * (q.v - p.v)};
      }
    }
    for (int i = 0; i < m; i++) poly[i] = out[i];
    return m;
  }
};

inline HostTables make_host_tables() {
  HostTables t;
  const int G = DIRMAP_G;
  t.cell.assign((size_t)6 * DIRMAP_CELLS, 0);
  // ---- pass 1 (parallel over hull x face): the candidate set of every cell, ascending ids, flat storage per task
  struct Task { std::vector<unsigned short> ids; std::vector<unsigned int> off; };
  std::vector<Task> tasks(36);
  auto run_task = [&](int h, int face) {
    Task& T = tasks[h * 6 + face];
    T.off.assign((size_t)G * G + 1, 0);
    T.ids.reserve((size_t)G * G * 2);
    const int v0 = UR5E_HULL_OFFSET[h], v1 = UR5E_HULL_OFFSET[h + 1];
    std::vector<int> mark(v1 - v0, -1), queue, parent;
    const CubeFace cf = cube_face(face);  // the hardware's cube-map conventions (urgym_device.h dirmap_cell)
    const int axis = cf.axis, au = cf.au, av = cf.av;
    const double s = cf.s, su = cf.su, sv = cf.sv;
    int cur = -1, row_start = -1, stamp = 0;
    for (int iv = 0; iv < G; iv++) {
      cur = row_start;
      for (int iu = 0; iu < G; iu++) {
        const double ulo = 2.0 * iu / G - 1.0 - CellCandidates::MARGIN, uhi = 2.0 * (iu + 1) / G - 1.0 + CellCandidates::MARGIN;
        const double vlo = 2.0 * iv / G - 1.0 - CellCandidates::MARGIN, vhi = 2.0 * (iv + 1) / G - 1.0 + CellCandidates::MARGIN;
        double d[3];
        d[axis] = s; d[au] = su * 0.5 * (ulo + uhi); d[av] = sv * 0.5 * (vlo + vhi);
        auto value = [&](int k) { return (UR5E_HULL_VERTS[k][0] * d[0] + UR5E_HULL_VERTS[k][1] * d[1]) + UR5E_HULL_VERTS[k][2] * d[2]; };
        if (cur < 0) {  // first cell of the face: scan
          cur = v0;
          for (int k = v0; k < v1; k++) if (value(k) > value(cur)) cur = k;
        }
        for (;;) {  // support vertex of the cell's centre: steepest ascent from the neighbouring cell's
          int nxt = cur;
          double bv = value(cur);
          for (int e = UR5E_ADJ_OFFSET[cur]; e < UR5E_ADJ_OFFSET[cur + 1]; e++) {
            const double val = value(UR5E_ADJ_INDEX[e]);
            if (val > bv) { bv = val; nxt = UR5E_ADJ_INDEX[e]; }
          }
          if (nxt == cur) break;
          cur = nxt;
        }
        if (iu == 0) row_start = cur;
        // half-plane of "p is at least as good as n" in the (u, v) plane of this face
        auto coef = [&](int p, int n, double& a, double& b, double& c) {
          const double e[3] = {UR5E_HULL_VERTS[p][0] - UR5E_HULL_VERTS[n][0], UR5E_HULL_VERTS[p][1] - UR5E_HULL_VERTS[n][1], UR5E_HULL_VERTS[p][2] - UR5E_HULL_VERTS[n][2]};
          a = su * e[au]; b = sv * e[av]; c = s * e[axis];
        };
        auto meets = [&](int p, int from) -> bool {
          double a, b, c;
          if (from >= 0) {  // cheap reject: the vertex the search came from beats p on the whole square
            coef(p, from, a, b, c);
            const double E = CellCandidates::EPS;
            if (a * ulo + b * vlo + c < -E && a * uhi + b * vlo + c < -E && a * ulo + b * vhi + c < -E && a * uhi + b * vhi + c < -E) return false;
          }
          CellCandidates::P2 poly[40] = {{ulo, vlo}, {uhi, vlo}, {uhi, vhi}, {ulo, vhi}};
          int np = 4;
          for (int e = UR5E_ADJ_OFFSET[p]; e < UR5E_ADJ_OFFSET[p + 1] && np > 0; e++) {
            coef(p, UR5E_ADJ_INDEX[e], a, b, c);
            np = CellCandidates::clip(poly, np, a, b, c);
          }
          return np > 0;
        };
        stamp++;
        queue.assign(1, cur);
        parent.assign(1, -1);
        mark[cur - v0] = stamp;
        const size_t first = T.ids.size();
        for (size_t qi = 0; qi < queue.size(); qi++) {
          const int p = queue[qi];
          if (!meets(p, parent[qi])) continue;
          T.ids.push_back((unsigned short)p);
          for (int e = UR5E_ADJ_OFFSET[p]; e < UR5E_ADJ_OFFSET[p + 1]; e++) {
            const int n = UR5E_ADJ_INDEX[e];
            if (mark[n - v0] != stamp) { mark[n - v0] = stamp; queue.push_back(n); parent.push_back(p); }
          }
        }
        if (T.ids.size() == first) T.ids.push_back((unsigned short)cur);  // (the centre's support vertex always meets its own cell)
        std::sort(T.ids.begin() + first, T.ids.end());
        T.off[(size_t)iv * G + iu + 1] = (unsigned int)T.ids.size();
      }
    }
  };
  {
    unsigned hw = std::thread::hardware_concurrency();
    const int workers = (int)std::max(1u, std::min(hw ? hw : 1u, 12u));
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; w++)
      pool.emplace_back([&, w]() { for (int k = w; k < 36; k += workers) run_task(k / 6, k % 6); });
    for (auto& th : pool) th.join();
  }
  // ---- pass 2 (sequential, deterministic): one record chain per DISTINCT candidate set, candidates by descending id, four per record
  auto set_slot = [&](CandRec& r, int j, int v) {
    double* xs = &r.x[0].a; double* ys = &r.y[0].a; double* zs = &r.z[0].a;
    xs[j] = UR5E_HULL_VERTS[v][0]; ys[j] = UR5E_HULL_VERTS[v][1]; zs[j] = UR5E_HULL_VERTS[v][2];
  };
  auto make_chain = [&](const unsigned short* ids, int n) -> int {  // ids ascending
    const int head = (int)t.recs.size();
    int rec = -1;
    for (int base = 0; base < n; base += 4) {
      t.recs.push_back(CandRec{});
      if (rec >= 0) t.recs[rec].next = (int)t.recs.size() - 1;
      rec = (int)t.recs.size() - 1;
      t.recs[rec].next = -1;
      t.recs[rec].pad[0] = t.recs[rec].pad[1] = t.recs[rec].pad[2] = 0;
      for (int j = 0; j < 4; j++) set_slot(t.recs[rec], j, ids[n - 1 - std::min(base + j, n - 1)]);  // descending
    }
    return head;
  };
  std::vector<int> single(UR5E_NUM_HULL_VERTS, -1);
  std::map<std::vector<unsigned short>, int> chains;
  for (int h = 0; h < 6; h++)
    for (int face = 0; face < 6; face++) {
      const Task& T = tasks[h * 6 + face];
      for (int c = 0; c < G * G; c++) {
        const unsigned short* ids = T.ids.data() + T.off[c];
        const int n = (int)(T.off[c + 1] - T.off[c]);
        int head;
        if (n == 1) {
          if (single[ids[0]] < 0) single[ids[0]] = make_chain(ids, 1);
          head = single[ids[0]];
        } else {
          std::vector<unsigned short> key(ids, ids + n);
          auto it = chains.find(key);
          if (it == chains.end()) it = chains.emplace(std::move(key), make_chain(ids, n)).first;
          head = it->second;
        }
        if (head > 65535) { t.ok = false; head = 0; }
        t.cell[(size_t)h * DIRMAP_CELLS + (size_t)face * G * G + c] = (unsigned short)head;
        t.cells_by_candidates[n <= 4 ? n - 1 : (n <= 8 ? 4 : 5)]++;
        t.longest_list = std::max(t.longest_list, n);
      }
    }
  return t;
}

// built once per process (tens of milliseconds on a many-core host; every handle uploads its own device copy)
inline const HostTables& build_host_tables() {
  static const HostTables tabs = make_host_tables();
  return tabs;
}

}  // namespace urgym
