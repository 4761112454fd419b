// oracle/ref_em_shim.cpp — TEST INFRASTRUCTURE.  A C entry point around the reference's single-threaded EM update, compiled from where the source
// lies under /root/reference (never copied) into oracle/_ref/libem_ref.so by oracle/Makefile:
//   src/inference/EMUtils.cpp   EMUpdate_ (the update the bootstrap replicates run, CollapsedEMOptimizer.cpp:380-470), truncateCountVector
// SalmonUtils.hpp (incLoop) and Transcript.hpp are stood in for by oracle/_stub/em.  Pins the checker's EM step (oracle.cpp em_step, use_vbem = 0) and
// through it the HIP kernels that are bit-exact with the checker — tests/test_em_pin.py.
#include "salmon/internal/inference/EMUtils.hpp"
#include <cstdint>
#include <vector>
extern "C" {
// CSR in (off[E + 1], tid[L], combined weights cw[L], count[E]); one EMUpdate_ from alpha_in into a zeroed alpha_out
void ref_em_update(uint64_t E, const uint64_t* off, const uint32_t* tid, const double* cw, const uint64_t* count, uint32_t M, const double* alpha_in, double* alpha_out) {
  std::vector<std::vector<uint32_t>> labels(E); std::vector<std::vector<double>> weights(E); std::vector<uint64_t> counts(count, count + E);
  for (uint64_t c = 0; c < E; ++c) { labels[c].assign(tid + off[c], tid + off[c + 1]); weights[c].assign(cw + off[c], cw + off[c + 1]); }
  std::vector<double> in(alpha_in, alpha_in + M), out(M, 0.0);
  EMUpdate_(labels, weights, counts, in, out);
  for (uint32_t i = 0; i < M; ++i) alpha_out[i] = out[i];
}
double ref_truncate(double* alphas, uint32_t M, double cutoff) { std::vector<double> a(alphas, alphas + M); const double s = truncateCountVector(a, cutoff); for (uint32_t i = 0; i < M; ++i) alphas[i] = a[i]; return s; }
}
