// oracle/ref_posbias_shim.cpp — TEST INFRASTRUCTURE.  C entry points around the reference's positional-bias model, compiled from where the
// sources lie under /root/reference (never copied) into oracle/_ref/libposbias_ref.so by oracle/Makefile:
//   src/model/SimplePosBias.cpp + include/salmon/internal/model/SimplePosBias.hpp   addMass / finalize / projectWeights
//   include/salmon/vendor/upstream/misc/spline.h                                     tk::spline (header-only)
// They pin the checker's restatement (oracle.cpp: pos_bin, pos_finalize, pos_spline_build / _eval) — tests/test_posbias_pin.py.
#include "salmon/internal/model/SimplePosBias.hpp"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <vector>
extern "C" {
// tk::spline through (xs, ys) with the constructor's defaults (natural cubic), evaluated at q
void ref_spline_eval(const double* xs, const double* ys, int n, const double* q, int nq, double* out) {
  tk::spline s(std::vector<double>(xs, xs + n), std::vector<double>(ys, ys + n));
  for (int i = 0; i < nq; ++i) out[i] = s(q[i]);
}
// a model whose bins hold exp(logmass[b]) + 1 (every bin starts at LOG_1): finalize(), then projectWeights over `len` positions;
// norm20 receives masses_ as writeBinary serialises them
void ref_pos_project(const double* logmass20, int32_t len, double* out, double* norm20) {
  SimplePosBias m(20, true);
  for (int b = 0; b < 20; ++b) m.addMass(b, logmass20[b]);
  m.finalize();
  std::vector<double> w((size_t)len); m.projectWeights(w); memcpy(out, w.data(), (size_t)len * 8);
  std::ostringstream os; m.writeBinary(os); const std::string s = os.str(); memcpy(norm20, s.data() + 4, 160);
}
// the bin addMass(pos, length, .) picks: a huge mass lands in exactly one bin
int ref_pos_bin(int32_t pos, int32_t length) {
  SimplePosBias m(20, true); m.addMass(pos, length, std::log(1e12)); m.finalize();
  std::ostringstream os; m.writeBinary(os); const std::string s = os.str(); double v[20]; memcpy(v, s.data() + 4, 160);
  int best = 0; for (int b = 1; b < 20; ++b) if (v[b] > v[best]) best = b; return best;
}
}
