// Prototype / statistics for the exact support lookup ("candidate lists per direction cell"): for every cell of the cube map of
// directions, the set of hull vertices that are the support vertex for SOME direction of the (slightly inflated) cell.  The device
// then needs no hill climb: the arg-max over the cell's candidates is the arg-max over the hull.
//   g++ -O2 -std=c++17 -o /tmp/gauss_stats tools/diag/gauss_stats.cpp && /tmp/gauss_stats [G]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#include "../../data/ur5e_model.h"

struct P2 { double u, v; };

// clip convex polygon by a*u + b*v + c >= -eps
static void clip(std::vector<P2>& poly, double a, double b, double c, double eps) {
  std::vector<P2> out;
  const size_t n = poly.size();
  for (size_t i = 0; i < n; i++) {
    const P2 p = poly[i], q = poly[(i + 1) % n];
    const double fp = a * p.u + b * p.v + c + eps, fq = a * q.u + b * q.v + c + eps;
    if (fp >= 0) out.push_back(p);
    if ((fp >= 0) != (fq >= 0)) {
      const double t = fp / (fp - fq);
      out.push_back(P2{p.u + t * (q.u - p.u), p.v + t * (q.v - p.v)});
    }
  }
  poly.swap(out);
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 128;
  const double margin = 1e-5, eps = 1e-11;
  auto t0 = std::chrono::steady_clock::now();
  long total_cells = 0;
  std::map<std::vector<int>, int> distinct;
  std::vector<long> hist(64, 0);
  std::vector<std::vector<int>> cellsets;  // for validation
  std::vector<int> cellhull;
  for (int h = 0; h < 6; h++) {
    const int v0 = UR5E_HULL_OFFSET[h], v1 = UR5E_HULL_OFFSET[h + 1];
    std::vector<int> mark(v1 - v0, -1);
    int stamp = 0;
    for (int face = 0; face < 6; face++) {
      const int axis = face / 2;
      const double s = (face & 1) ? -1.0 : 1.0;
      const int au = (axis + 1) % 3, av = (axis + 2) % 3;
      int cur = -1;
      for (int iv = 0; iv < G; iv++)
        for (int iu = 0; iu < G; iu++) {
          const double ulo = 2.0 * iu / G - 1.0 - margin, uhi = 2.0 * (iu + 1) / G - 1.0 + margin;
          const double vlo = 2.0 * iv / G - 1.0 - margin, vhi = 2.0 * (iv + 1) / G - 1.0 + margin;
          double d[3];
          d[axis] = s; d[au] = 0.5 * (ulo + uhi); d[av] = 0.5 * (vlo + vhi);
          auto val = [&](int k) { return UR5E_HULL_VERTS[k][0] * d[0] + UR5E_HULL_VERTS[k][1] * d[1] + UR5E_HULL_VERTS[k][2] * d[2]; };
          if (cur < 0) {
            cur = v0;
            for (int k = v0; k < v1; k++) if (val(k) > val(cur)) cur = k;
          }
          for (;;) {
            int nxt = cur; double bv = val(cur);
            for (int e = UR5E_ADJ_OFFSET[cur]; e < UR5E_ADJ_OFFSET[cur + 1]; e++) { const double x = val(UR5E_ADJ_INDEX[e]); if (x > bv) { bv = x; nxt = UR5E_ADJ_INDEX[e]; } }
            if (nxt == cur) break;
            cur = nxt;
          }
          // BFS over the hull graph from the centre's support vertex: p belongs to the cell iff its (eps-relaxed) normal cone meets the square
          stamp++;
          std::vector<int> set, queue{cur};
          mark[cur - v0] = stamp;
          auto feasible = [&](int p, int parent) -> bool {
            // constraints: d . (p - n) >= -eps for every neighbour n, d = (s, u, v) -> a u + b v + c >= -eps
            auto coef = [&](int n, double& a, double& b, double& c) {
              const double e0 = UR5E_HULL_VERTS[p][0] - UR5E_HULL_VERTS[n][0], e1 = UR5E_HULL_VERTS[p][1] - UR5E_HULL_VERTS[n][1], e2 = UR5E_HULL_VERTS[p][2] - UR5E_HULL_VERTS[n][2];
              const double e[3] = {e0, e1, e2};
              a = e[au]; b = e[av]; c = s * e[axis];
            };
            auto allout = [&](double a, double b, double c) {
              return a * ulo + b * vlo + c < -eps && a * uhi + b * vlo + c < -eps && a * ulo + b * vhi + c < -eps && a * uhi + b * vhi + c < -eps;
            };
            double a, b, c;
            if (parent >= 0) { coef(parent, a, b, c); if (allout(a, b, c)) return false; }
            std::vector<P2> poly{{ulo, vlo}, {uhi, vlo}, {uhi, vhi}, {ulo, vhi}};
            for (int e = UR5E_ADJ_OFFSET[p]; e < UR5E_ADJ_OFFSET[p + 1] && !poly.empty(); e++) {
              coef(UR5E_ADJ_INDEX[e], a, b, c);
              clip(poly, a, b, c, eps);
            }
            return !poly.empty();
          };
          std::vector<int> parent{-1};
          for (size_t qi = 0; qi < queue.size(); qi++) {
            const int p = queue[qi];
            if (!feasible(p, parent[qi])) continue;
            set.push_back(p);
            for (int e = UR5E_ADJ_OFFSET[p]; e < UR5E_ADJ_OFFSET[p + 1]; e++) {
              const int n = UR5E_ADJ_INDEX[e];
              if (mark[n - v0] != stamp) { mark[n - v0] = stamp; queue.push_back(n); parent.push_back(p); }
            }
          }
          if (set.empty()) set.push_back(cur);  // (cannot happen: the centre's support vertex is feasible)
          std::sort(set.begin(), set.end());
          hist[std::min<size_t>(set.size(), 63)]++;
          distinct[set]++;
          total_cells++;
          cellsets.push_back(set);
          cellhull.push_back(h);
        }
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  printf("G = %d: %ld cells, build %.2f s, distinct candidate sets %zu\n", G, total_cells, std::chrono::duration<double>(t1 - t0).count(), distinct.size());
  long acc = 0;
  for (int k = 1; k < 64; k++) if (hist[k]) { acc += hist[k]; printf("  %2d%s candidates: %8ld cells (%.2f %%, cumulative %.2f %%)\n", k, k == 63 ? "+" : "", hist[k], 100.0 * hist[k] / total_cells, 100.0 * acc / total_cells); }
  // records needed if a record holds R candidates
  for (int R : {2, 3, 4, 6, 8}) {
    long recs = 0;
    for (auto& kv : distinct) recs += ((long)kv.first.size() + R - 1) / R;
    printf("  records of %d candidates: %ld records\n", R, recs);
  }
  // validation: random directions, the scan's vertex must be in the cell's set (cell chosen with float32 arithmetic like the device)
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  long bad = 0, tested = 0;
  for (int h = 0; h < 6; h++) {
    const int v0 = UR5E_HULL_OFFSET[h], v1 = UR5E_HULL_OFFSET[h + 1];
    for (int it = 0; it < 300000; it++) {
      double d[3] = {nd(rng), nd(rng), nd(rng)};
      if (it % 3 == 0) {  // adversarial: the normal of a hull edge / near-face direction: difference-orthogonal directions
        const int p = v0 + (int)(rng() % (v1 - v0));
        const int e = UR5E_ADJ_OFFSET[p] + (int)(rng() % (UR5E_ADJ_OFFSET[p + 1] - UR5E_ADJ_OFFSET[p]));
        const int n = UR5E_ADJ_INDEX[e];
        const double ex = UR5E_HULL_VERTS[p][0] - UR5E_HULL_VERTS[n][0], ey = UR5E_HULL_VERTS[p][1] - UR5E_HULL_VERTS[n][1], ez = UR5E_HULL_VERTS[p][2] - UR5E_HULL_VERTS[n][2];
        const double k = (d[0] * ex + d[1] * ey + d[2] * ez) / (ex * ex + ey * ey + ez * ez);
        d[0] -= k * ex; d[1] -= k * ey; d[2] -= k * ez;  // now p and n tie (up to rounding)
      }
      int best = v0;
      double bv = -1e300;
      for (int k = v0; k < v1; k++) { const double x = (UR5E_HULL_VERTS[k][0] * d[0] + UR5E_HULL_VERTS[k][1] * d[1]) + UR5E_HULL_VERTS[k][2] * d[2]; if (x > bv) { bv = x; best = k; } }
      const float x = (float)d[0], y = (float)d[1], z = (float)d[2];
      const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
      const int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
      const float m = axis == 0 ? x : (axis == 1 ? y : z), u = axis == 0 ? y : (axis == 1 ? z : x), v = axis == 0 ? z : (axis == 1 ? x : y);
      float am = fabsf(m); if (!(am > 0.0f)) am = 1.0f;
      const float inv = 1.0f / am;
      int iu = (int)((u * inv + 1.0f) * (0.5f * G)), iv = (int)((v * inv + 1.0f) * (0.5f * G));
      iu = iu < 0 ? 0 : (iu > G - 1 ? G - 1 : iu); iv = iv < 0 ? 0 : (iv > G - 1 ? G - 1 : iv);
      const int face = axis * 2 + (m < 0.0f ? 1 : 0);
      const size_t cell = (size_t)h * 6 * G * G + ((size_t)face * G + iv) * G + iu;
      const auto& set = cellsets[cell];
      tested++;
      if (!std::binary_search(set.begin(), set.end(), best)) bad++;
    }
  }
  printf("validation: %ld directions, scan's vertex missing from the cell's set: %ld\n", tested, bad);
  return 0;
}
