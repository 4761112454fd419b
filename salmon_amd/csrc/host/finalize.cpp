// host/finalize.cpp — the once-per-run host steps between the online phase and EM, and the output
// writers that define the drop-in file contract.
//   sq_normalize_alphas   = salmon::utils::normalizeAlphas (reference src/util/SalmonUtils.cpp:461-529)
//                           + TranscriptCluster::projectToPolytope (include/salmon/internal/quant/TranscriptCluster.hpp:46-102)
//   sq_write_quant_sf     = GZipWriter::writeAbundances (src/output/GZipWriter.cpp:684-739)
//   sq_write_eq_classes   = GZipWriter::writeEquivCounts (src/output/GZipWriter.cpp:64-168), gzip text
// The reference keeps clusters in a union-find that is merged read by read under a mutex
// (ClusterForest.hpp:30-59); the clusters are exactly the connected components of the eq-class
// labels, so they are rebuilt here from the final table (members visited in ascending transcript id).
#include "index.h"
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <map>
#include <numeric>

namespace {
struct DSU {
  std::vector<uint32_t> p;
  explicit DSU(uint32_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
  uint32_t root(uint32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
  void join(uint32_t a, uint32_t b) { a = root(a); b = root(b); if (a == b) return; if (a < b) p[b] = a; else p[a] = b; }
};
}  // namespace

extern "C" int sq_normalize_alphas(uint32_t M, const sq_eq_table* eq, const double* log_mass, const uint64_t* uniq, const uint64_t* total, double* projected) {
  if (!eq || !log_mass || !uniq || !total || !projected) { sq_set_error("sq_normalize_alphas: bad arguments"); return SQ_ERR_ARG; }
  DSU d(M);
  for (uint64_t c = 0; c < eq->num_classes; ++c) { const uint64_t a = eq->off[c], b = eq->off[c + 1]; for (uint64_t i = a + 1; i < b; ++i) d.join(eq->tid[a], eq->tid[i]); }
  std::vector<double> hits(M, 0.0);
  for (uint64_t c = 0; c < eq->num_classes; ++c) if (eq->off[c + 1] > eq->off[c]) hits[d.root(eq->tid[eq->off[c]])] += (double)eq->count[c];
  // bucket members by root (counting sort keeps ascending tid inside each cluster)
  std::vector<uint32_t> start(M + 1, 0), order(M);
  for (uint32_t t = 0; t < M; ++t) start[d.root(t) + 1]++;
  for (uint32_t r = 0; r < M; ++r) start[r + 1] += start[r];
  { std::vector<uint32_t> cur(start.begin(), start.end() - 1); for (uint32_t t = 0; t < M; ++t) order[cur[d.root(t)]++] = t; }
  std::vector<uint8_t> bound;
  for (uint32_t r = 0; r < M; ++r) {
    const uint32_t lo = start[r], hi = start[r + 1]; if (lo == hi) continue;
    double clusterMass = SQ_LOG_0;
    for (uint32_t i = lo; i < hi; ++i) clusterMass = sq_log_add(clusterMass, log_mass[order[i]]);     // SalmonUtils.cpp:483-490
    const double clusterCount = hits[r]; const double logCount = clusterCount > 0 ? sq_log(clusterCount) : SQ_LOG_0;
    bool project = false;
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t t = order[i];
      if (log_mass[t] == SQ_LOG_0) { projected[t] = 0.0; continue; }                                      // :500-501
      projected[t] = clusterCount > 0 ? sq_exp((log_mass[t] - clusterMass) + logCount) : 0.0;           // :503-504
      project |= projected[t] > (double)total[t] || projected[t] < (double)uniq[t];                       // :505-507
    }
    if (hi - lo > 1 && project) {  // projectToPolytope
      bound.assign(hi - lo, 0);
      for (int round = 0; round <= 5000; ++round) {
        double freeSum = 0.0, fixedSum = 0.0;
        for (uint32_t i = lo; i < hi; ++i) {
          const uint32_t t = order[i]; double& pc = projected[t];
          if (pc > (double)total[t]) { pc = (double)total[t]; bound[i - lo] = 1; }
          else if (pc < (double)uniq[t]) { pc = (double)uniq[t]; bound[i - lo] = 1; }
          (bound[i - lo] ? fixedSum : freeSum) += pc;
        }
        if (std::fabs(freeSum + fixedSum - clusterCount) <= 0.375e-10) break;  // approxEqual, SalmonMath.hpp:49-52
        if (freeSum == 0) { std::fill(bound.begin(), bound.end(), 0); freeSum = fixedSum; fixedSum = 0; }
        const double scale = (clusterCount - fixedSum) / freeSum;
        for (uint32_t i = lo; i < hi; ++i) if (!bound[i - lo]) projected[order[i]] *= scale;
      }
    }
  }
  return SQ_OK;
}

// quant.sf: Name Length EffectiveLength TPM NumReads (GZipWriter.cpp:698-736)
extern "C" int sq_write_quant_sf(const char* path, const sq_index* idx, const double* eff_len, const double* num_reads, double num_mapped_frags) {
  if (!path || !idx || !eff_len || !num_reads) { sq_set_error("sq_write_quant_sf: bad arguments"); return SQ_ERR_ARG; }
  FILE* f = fopen(path, "w"); if (!f) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  const uint32_t M = (uint32_t)idx->names.size();
  if (!(num_mapped_frags > 0)) { num_mapped_frags = 0; for (uint32_t i = 0; i < M; ++i) num_mapped_frags += num_reads[i]; }   // explicitSum (:704-708)
  double denom = 0.0;
  for (uint32_t i = 0; i < M; ++i) denom += (num_reads[i] / num_mapped_frags) / eff_len[i];                                       // :718-722
  fprintf(f, "Name\tLength\tEffectiveLength\tTPM\tNumReads\n");
  for (uint32_t i = 0; i < M; ++i) {
    double npm = num_reads[i] / num_mapped_frags; double tpm = denom > 0 ? ((npm / eff_len[i]) / denom) * 1000000.0 : 0.0;
    fprintf(f, "%s\t%u\t%.3f\t%f\t%.3f\n", idx->names[i].c_str(), idx->ref_clen[i], eff_len[i], tpm, num_reads[i]);              // sigDigits = 3
  }
  fclose(f);
  return SQ_OK;
}

// aux_info/eq_classes.txt.gz (doc/source/file_formats.rst:173-253). with_weights = --dumpEqWeights;
// otherwise range-factorised classes are collapsed to transcript sets (GZipWriter.cpp:89-114).
extern "C" int sq_write_eq_classes(const char* path, const sq_index* idx, const sq_eq_table* eq, int with_weights) {
  if (!path || !idx || !eq) { sq_set_error("sq_write_eq_classes: bad arguments"); return SQ_ERR_ARG; }
  gzFile g = gzopen(path, "wb"); if (!g) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  const uint32_t M = (uint32_t)idx->names.size();
  if (with_weights) {
    gzprintf(g, "%u\n%llu\n", M, (unsigned long long)eq->num_classes);
    for (auto& n : idx->names) gzprintf(g, "%s\n", n.c_str());
    for (uint64_t c = 0; c < eq->num_classes; ++c) {
      const uint64_t a = eq->off[c], b = eq->off[c + 1];
      gzprintf(g, "%llu\t", (unsigned long long)(b - a));
      for (uint64_t i = a; i < b; ++i) gzprintf(g, "%u\t", eq->tid[i]);
      for (uint64_t i = a; i < b; ++i) gzprintf(g, "%.17g\t", eq->w[i]);
      gzprintf(g, "%llu\n", (unsigned long long)eq->count[c]);
    }
  } else {
    std::map<std::vector<uint32_t>, uint64_t> col;
    for (uint64_t c = 0; c < eq->num_classes; ++c) { std::vector<uint32_t> k(eq->tid + eq->off[c], eq->tid + eq->off[c + 1]); col[k] += eq->count[c]; }
    gzprintf(g, "%u\n%zu\n", M, col.size());
    for (auto& n : idx->names) gzprintf(g, "%s\n", n.c_str());
    for (auto& kv : col) { gzprintf(g, "%zu\t", kv.first.size()); for (uint32_t t : kv.first) gzprintf(g, "%u\t", t); gzprintf(g, "%llu\n", (unsigned long long)kv.second); }
  }
  gzclose(g);
  return SQ_OK;
}
