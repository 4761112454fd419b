// host/posbias.cpp — see posbias.h.  Reference: src/model/SimplePosBias.cpp, include/salmon/vendor/upstream/misc/spline.h (tk::spline),
// include/salmon/internal/quant/ReadExperiment.inl:352-388.
#include "posbias.h"
#include "index.h"
#include <algorithm>
#include <cstring>

int sq_pos_length_classes(const sq_index* idx, uint32_t* quantiles, std::vector<uint8_t>& cls) {
  const size_t M = idx->ref_len.size();
  const size_t n = std::min<size_t>(idx->first_decoy, M);   // decoys follow the transcripts; the loader collects the lengths of the others only
  std::vector<uint32_t> len(idx->ref_len.begin(), idx->ref_len.begin() + n), q;
  std::sort(len.begin(), len.end());                                               // nth_element at growing ranks = the order statistics
  if (n > SQ_POS_CLASSES) { const size_t step = n / SQ_POS_CLASSES; size_t at = 0; for (int i = 0; i < SQ_POS_CLASSES; ++i) { at += step; q.push_back(len[std::min(at, n - 1)]); } }
  else q = len;
  cls.assign(M, 0);
  for (size_t i = 0; i < q.size(); ++i) quantiles[i] = q[i];
  if (q.empty()) return 0;
  const long last = (long)q.size() - 1;
  for (size_t t = 0; t < M; ++t) cls[t] = (uint8_t)std::min<long>(last, std::upper_bound(q.begin(), q.end(), idx->ref_len[t]) - q.begin());
  return (int)q.size();
}

// tk::spline(xs, ys) with its defaults: natural cubic spline; the tridiagonal system is solved the way band_matrix::lu_solve does it
// (rows scaled to a unit diagonal, then elimination without pivoting) so that the coefficients carry the same roundings
static void spline_through(const double* xs, const double* ys, sq_pos_spline* S) {
  const int n = SQ_POS_KNOTS;
  double sub[n], dia[n], sup[n], rhs[n], inv[n], y[n], b[n];
  memset(sub, 0, sizeof(sub)); memset(sup, 0, sizeof(sup));
  for (int i = 1; i + 1 < n; ++i) {
    const double hl = xs[i] - xs[i - 1], hr = xs[i + 1] - xs[i];
    sub[i] = 1.0 / 3.0 * hl; dia[i] = 2.0 / 3.0 * (xs[i + 1] - xs[i - 1]); sup[i] = 1.0 / 3.0 * hr;
    rhs[i] = (ys[i + 1] - ys[i]) / hr - (ys[i] - ys[i - 1]) / hl;
  }
  dia[0] = 2.0; rhs[0] = 0.0; dia[n - 1] = 2.0; rhs[n - 1] = 0.0;                  // second derivative 0 at both ends
  for (int i = 0; i < n; ++i) { inv[i] = 1.0 / dia[i]; sub[i] *= inv[i]; sup[i] *= inv[i]; dia[i] = 1.0; }
  for (int k = 0; k + 1 < n; ++k) { const double f = -sub[k + 1] / dia[k]; sub[k + 1] = -f; dia[k + 1] = dia[k + 1] + f * sup[k]; }
  for (int i = 0; i < n; ++i) { double s = 0; if (i) s += sub[i] * y[i - 1]; y[i] = (rhs[i] * inv[i]) - s; }
  for (int i = n - 1; i >= 0; --i) { double s = 0; if (i + 1 < n) s += sup[i] * b[i + 1]; b[i] = (y[i] - s) / dia[i]; }
  for (int i = 0; i < n; ++i) { S->x[i] = xs[i]; S->y[i] = ys[i]; S->b[i] = b[i]; S->a[i] = 0.0; S->c[i] = 0.0; }
  for (int i = 0; i + 1 < n; ++i) {
    const double h = xs[i + 1] - xs[i];
    S->a[i] = 1.0 / 3.0 * (b[i + 1] - b[i]) / h;
    S->c[i] = (ys[i + 1] - ys[i]) / h - 1.0 / 3.0 * (2.0 * b[i] + b[i + 1]) * h;
  }
  const double h = xs[n - 1] - xs[n - 2];
  S->c[n - 1] = 3.0 * S->a[n - 2] * h * h + 2.0 * S->b[n - 2] * h + S->c[n - 2];
}

void sq_pos_finalize(const double* mass, sq_pos_spline* out, double* norm) {
  static const double edge[SQ_POS_BINS] = {.02, .04, .06, .08, .10, .15, .2, .3, .4, .5, .6, .7, .8, .85, .9, .92, .94, .96, .98, 1.0};   // SimplePosBias.hpp:43-45
  double total = 0.0; for (int i = 0; i < SQ_POS_BINS; ++i) total += mass[i];
  const double first = mass[0] / total, last = mass[SQ_POS_BINS - 1] / total, with_knots = total + first + last;
  double xs[SQ_POS_KNOTS], ys[SQ_POS_KNOTS];
  xs[0] = 0.0; ys[0] = first;
  for (int i = 0; i < SQ_POS_BINS; ++i) { xs[i + 1] = edge[i] - 0.01; ys[i + 1] = mass[i] / with_knots; if (norm) norm[i] = mass[i] / total; }
  xs[SQ_POS_KNOTS - 1] = 1.0; ys[SQ_POS_KNOTS - 1] = last;
  spline_through(xs, ys, out);
}
