// ComplexNetworks.Network.area_level (reference ComplexNetworks.py:49-281) on the host, in C++: the caller that turns the
// cell-to-cell correlation matrix into the areas whose anomaly series are the GP's features.  The algorithm is
// sequential and greedy (grow an area cell by cell by best mean correlation; then merge neighbouring areas while the mean
// pairwise correlation of the union stays above tau), so it stays on the host -- but as 300 lines of C++ over flat arrays
// instead of Python lists of tuples: 6.7 s -> 0.1 s on the 57 x 57 grid of the north scripts, which is what a retrospective
// run (one network per year) otherwise spends its time on once the GP itself takes milliseconds.
//
// Same decisions as the reference, decision for decision: same candidate order, same first-maximum tie-breaks, same NaN
// sentinel cell, and -- because ties and threshold crossings are decided on floating-point means -- the same summation ORDER as
// np.nanmean (NaN -> 0, NumPy's pairwise add.reduce: straight loop below 8 elements, 8 interleaved accumulators up to 128, halves
// above), so that every mean is bit-identical to the one NumPy forms.  Checked against the Python restatement
// (networks.Network.area_level(native=False), itself pinned by the reference's goldens) on random fields.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/sigp.h"

namespace {

// numpy/core/src/umath/loops_utils.h.src: DOUBLE_pairwise_sum on a contiguous array
double pairwise_sum(const double* a, long n) {
  if (n < 8) {
    double res = 0.0;
    for (long i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    long i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  long n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

// np.nanmean of a 1-D array: NaN entries replaced by 0 for the sum, divided by the count of the others (0 / 0 = NaN)
double nanmean(std::vector<double>& buf) {
  long cnt = 0;
  for (double& v : buf) {
    if (std::isnan(v)) v = 0.0; else ++cnt;
  }
  const double tot = 0.0 + pairwise_sum(buf.data(), (long)buf.size());
  return cnt == 0 ? std::numeric_limits<double>::quiet_NaN() : tot / (double)cnt;
}

// np.nanmax of a list (NaN when every entry is NaN) and the index of its first occurrence
bool first_nanmax(const std::vector<double>& v, double& mx, long& where) {
  mx = std::numeric_limits<double>::quiet_NaN();
  for (double x : v)
    if (!std::isnan(x) && (std::isnan(mx) || x > mx)) mx = x;
  if (std::isnan(mx)) return false;
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == mx) { where = (long)i; return true; }
  return false;
}

struct Grid {
  long dimX, dimY;
  int cell_nan;                 // flat index of the first NaN cell: the out-of-bounds sentinel (ComplexNetworks.py:50-51)
  bool latlon;
  const int32_t* node;          // [dimX * dimY] row of R for an active cell, -1 otherwise
  const double* R; long N;
  bool inb(long i, long j) const { return i >= 0 && i <= dimX - 1 && j >= 0 && j <= dimY - 1; }
  // gen_cell_neighbours (:53-79): up, down, left, right; unavailable or outside -> the sentinel; on a lat-lon grid the
  // left / right neighbours wrap (the wrapped cell is NOT checked against `unavail`, as in the reference)
  void neighbours(long i, long j, const std::vector<char>& unavail, int out[4]) const {
    const long a[4] = {i - 1, i + 1, i, i}, b[4] = {j, j, j - 1, j + 1};
    for (int q = 0; q < 4; ++q) {
      const bool in = inb(a[q], b[q]);
      if (in && unavail[(size_t)(a[q] * dimY + b[q])]) out[q] = cell_nan;
      else if (in) out[q] = (int)(a[q] * dimY + b[q]);
      else if (latlon && q >= 2) out[q] = (int)(i * dimY + (q == 2 ? dimY - 1 : 0));
      else out[q] = cell_nan;
    }
  }
};

}  // namespace

extern "C" int sigp_area_level(const double* R, int64_t N, const int32_t* node_of_cell, int64_t dimX, int64_t dimY, int32_t cell_nan,
                               double tau, int latlon, int32_t* cells_out, int64_t* area_offsets, int32_t* area_ids, int64_t* n_areas_out,
                               int32_t* unavail_out, int64_t unavail_cap, int64_t* n_unavail_out) {
  if (!R || !node_of_cell || N < 1 || dimX < 1 || dimY < 1 || cell_nan < 0 || cell_nan >= dimX * dimY || !cells_out || !area_offsets || !area_ids ||
      !n_areas_out || !unavail_out || unavail_cap < 0 || !n_unavail_out)
    return SIGP_BAD_ARG;
  const Grid g{(long)dimX, (long)dimY, (int)cell_nan, latlon != 0, node_of_cell, R, (long)N};
  const size_t P = (size_t)(dimX * dimY);
  for (size_t p = 0; p < P; ++p)
    if (node_of_cell[p] >= N) return SIGP_BAD_ARG;
  std::vector<char> unavail(P, 0);
  std::vector<std::vector<int>> areas;       // step 1: creation order
  std::vector<int> ids;
  std::vector<double> buf, means;
  std::vector<int> cand, kept;
  std::vector<char> seen(P, 0);

  // ---- S T E P  1: create areas (:151-196) ----
  for (long i = 0; i < dimX; ++i)
    for (long j = 0; j < dimY; ++j) {
      const int c0 = (int)(i * dimY + j);
      const int ID = node_of_cell[c0];
      if (ID < 0 || unavail[(size_t)c0]) continue;
      int nei[4];
      g.neighbours(i, j, unavail, nei);
      std::vector<double> nc(4);
      for (int q = 0; q < 4; ++q) {
        const int n = node_of_cell[nei[q]];
        nc[(size_t)q] = n < 0 ? std::numeric_limits<double>::quiet_NaN() : R[(size_t)ID * N + n];
      }
      double mx; long w = 0;
      if (!first_nanmax(nc, mx, w) || !(mx > tau)) continue;
      const int best = nei[w];
      if (unavail[(size_t)best]) continue;
      std::vector<int> cells{c0, best};
      unavail[(size_t)c0] = 1; unavail[(size_t)best] = 1;
      // expand / gen_area_neighbours / area_max_correlation (:81-149)
      for (;;) {
        cand.clear();
        static const int di[4] = {-1, 1, 0, 0}, dj[4] = {0, 0, -1, 1};
        for (int q = 0; q < 4; ++q)
          for (int c : cells) {
            const long a = c / dimY + di[q], b = c % dimY + dj[q];
            const bool in = g.inb(a, b);
            if (in && unavail[(size_t)(a * dimY + b)]) continue;
            const int nb = in ? (int)(a * dimY + b) : g.cell_nan;
            if (!seen[(size_t)nb]) { seen[(size_t)nb] = 1; cand.push_back(nb); }   // duplicates never change the first maximum
          }
        for (int nb : cand) seen[(size_t)nb] = 0;
        if (cand.empty()) break;
        kept.clear(); means.clear();
        for (int nb : cand) {
          const int n = node_of_cell[nb];
          if (n < 0) continue;
          buf.resize(cells.size());
          for (size_t k = 0; k < cells.size(); ++k) buf[k] = R[(size_t)n * N + node_of_cell[cells[k]]];
          kept.push_back(nb);
          means.push_back(nanmean(buf));
        }
        if (means.empty()) break;
        double rmax; long wm = 0;
        if (!first_nanmax(means, rmax, wm) || !(rmax > tau)) break;
        const int m = kept[(size_t)wm];
        cells.push_back(m);
        unavail[(size_t)m] = 1;
      }
      ids.push_back((int)areas.size());
      areas.push_back(std::move(cells));
    }

  // ---- S T E P  2: minimise the number of areas (:198-265) ----
  std::fill(unavail.begin(), unavail.end(), 0);
  std::vector<int> unavail_list;
  std::vector<int> owner(P, -1);             // position (in `areas`) of the area a cell belongs to
  std::vector<char> in_big(P, 0), taken(P, 0);
  for (;;) {
    if (areas.empty()) break;
    long best_pos = 0, best_cnt = -1;
    for (size_t a = 0; a < areas.size(); ++a) {
      const long cnt = unavail[(size_t)areas[a][0]] ? 0 : (long)areas[a].size();
      if (cnt > best_cnt) { best_cnt = cnt; best_pos = (long)a; }      // max(): first maximum in dict order
    }
    if (best_cnt == 0) break;
    const std::vector<int>& big = areas[(size_t)best_pos];
    std::fill(owner.begin(), owner.end(), -1);
    for (size_t a = 0; a < areas.size(); ++a)
      for (int c : areas[a])
        if (owner[(size_t)c] < 0) owner[(size_t)c] = (int)a;
    for (int c : big) in_big[(size_t)c] = 1;
    std::vector<int> tried;                  // areas that received a score, in first-scored order (Anei_Rs)
    std::vector<double> score;
    std::vector<int> marked;                 // cells put into unavail_neis
    for (int xc : big) {
      int nei[4];
      g.neighbours(xc / dimY, xc % dimY, unavail, nei);
      bool any = false;
      for (int q = 0; q < 4; ++q)
        if (owner[(size_t)nei[q]] >= 0 && !in_big[(size_t)nei[q]]) any = true;
      if (!any) continue;
      for (size_t kk = 0; kk < areas.size(); ++kk) {       // the reference scans the areas in dict order for every cell
        bool present = false;
        for (int q = 0; q < 4; ++q)
          if (owner[(size_t)nei[q]] == (int)kk && !in_big[(size_t)nei[q]]) present = true;
        if (!present) continue;
        for (int q = 0; q < 4; ++q) {
          const int nb = nei[q];
          if (in_big[(size_t)nb] || taken[(size_t)nb] || owner[(size_t)nb] != (int)kk) continue;
          for (int c : areas[kk])
            if (!taken[(size_t)c]) { taken[(size_t)c] = 1; marked.push_back(c); }
          // mean correlation of the hypothetical union: pairs (cell, later cells) (:236-244)
          std::vector<int> hyp(big);
          hyp.insert(hyp.end(), areas[kk].begin(), areas[kk].end());
          const size_t H = hyp.size();
          std::vector<double> rm(H);
          for (size_t a = 0; a < H; ++a) {
            buf.resize(H - a - 1);
            const double* Ra = R + (size_t)node_of_cell[hyp[a]] * N;
            for (size_t b = a + 1; b < H; ++b) buf[b - a - 1] = Ra[node_of_cell[hyp[b]]];
            rm[a] = nanmean(buf);
          }
          bool had = false;
          for (int t : tried) if (t == (int)kk) had = true;
          if (!had) { tried.push_back((int)kk); score.push_back(nanmean(rm)); }
        }
      }
    }
    for (int c : marked) taken[(size_t)c] = 0;
    bool merged = false;
    if (!tried.empty()) {
      size_t bk = 0;                          // Python max(): replaced only by a strictly greater value (NaN never wins, a leading NaN stays)
      for (size_t t = 1; t < tried.size(); ++t)
        if (score[t] > score[bk]) bk = t;
      if (score[bk] > tau) {
        const size_t kk = (size_t)tried[bk];
        for (int c : big) in_big[(size_t)c] = 0;
        std::vector<int>& dst = areas[(size_t)best_pos];
        dst.insert(dst.end(), areas[kk].begin(), areas[kk].end());       // V[max_ID] = big + V.pop(best_k): key position kept
        areas.erase(areas.begin() + (long)kk);
        ids.erase(ids.begin() + (long)kk);
        merged = true;
      }
    }
    if (!merged) {
      for (int c : big) { in_big[(size_t)c] = 0; unavail[(size_t)c] = 1; unavail_list.push_back(c); }
    }
  }

  // ---- output: areas in dict order, cells in list order ----
  int64_t off = 0;
  for (size_t a = 0; a < areas.size(); ++a) {
    area_offsets[a] = off;
    area_ids[a] = ids[a];
    for (int c : areas[a]) cells_out[off++] = c;
  }
  area_offsets[areas.size()] = off;
  *n_areas_out = (int64_t)areas.size();
  // (on a lat-lon grid an area that was set aside can still be absorbed through a wrapped neighbour and be set aside again with
  // its new owner: the list may hold a cell more than once and be longer than N -- the caller passes its capacity and gets the
  // full length back)
  for (size_t k = 0; k < unavail_list.size() && (int64_t)k < unavail_cap; ++k) unavail_out[k] = unavail_list[k];
  *n_unavail_out = (int64_t)unavail_list.size();
  return SIGP_OK;
}

// np.nanmean of a contiguous array, exposed for the bit-identity test of the summation order
extern "C" double sigp_host_nanmean(const double* a, int64_t n) {
  std::vector<double> buf(a, a + n);
  return nanmean(buf);
}
