// oracle/ref_polytope_shim.cpp — TEST INFRASTRUCTURE.  A C entry point around the reference's transcript clusters, compiled from where the headers lie
// under /root/reference (never copied) into oracle/_ref/libpolytope_ref.so by oracle/Makefile:
//   include/salmon/internal/quant/TranscriptCluster.hpp   projectToPolytope (:46-102), merge, members
//   include/salmon/internal/quant/ClusterForest.hpp        mergeClusters, updateCluster, getClusters
// boost::dynamic_bitset / boost::disjoint_sets / Transcript are stood in for by oracle/_stub/poly.  normalizeAlphas itself lives in SalmonUtils.cpp
// (2 300 lines with every dependency of the program), so its 40 lines (src/util/SalmonUtils.cpp:460-527) are restated around the reference's
// own cluster objects here.  Pins the checker's orc_normalize_alphas (row a14) — tests/test_polytope_pin.py.
#include "salmon/internal/model/Transcript.hpp"
#include "salmon/internal/quant/ClusterForest.hpp"
#include <cmath>
#include <cstdint>
#include <vector>
namespace { struct Frag { uint32_t t; uint32_t transcriptID() const { return t; } }; }
extern "C" void ref_normalize_alphas(uint32_t M, uint64_t E, const uint64_t* off, const uint32_t* tid, const uint64_t* count, const double* log_mass,
                                     const uint64_t* unique, const uint64_t* total, uint64_t num_mapped, double* projected_out, uint64_t* num_clusters_out) {
  using salmon::math::LOG_0;
  std::vector<Transcript> refs(M);
  for (uint32_t i = 0; i < M; ++i) { refs[i].logMass_ = std::isinf(log_mass[i]) ? LOG_0 : log_mass[i]; refs[i].uniq_ = unique[i]; refs[i].total_ = total[i]; }
  ClusterForest forest(M, refs);
  // what processMiniBatch does per alignment group (SalmonQuantify.cpp:1003-1008), once per class with the class's count
  for (uint64_t c = 0; c < E; ++c) {
    std::vector<Frag> fr; for (uint64_t i = off[c]; i < off[c + 1]; ++i) fr.push_back(Frag{tid[i]});
    forest.mergeClusters<Frag>(fr.begin(), fr.end());
    forest.updateCluster(fr.front().transcriptID(), (size_t)count[c], LOG_0, true);
  }
  // normalizeAlphas (src/util/SalmonUtils.cpp:460-527)
  auto clusters = forest.getClusters();
  for (auto cptr : clusters) {
    double logClusterMass = LOG_0; const double logClusterCount = std::log(static_cast<double>(cptr->numHits()));
    bool requiresProjection = false; auto& members = cptr->members(); size_t clusterSize = 0;
    for (auto t_id : members) { Transcript& t = refs[t_id]; t.uniqueCounts = t.uniqueCount(); t.totalCounts = t.totalCount(); logClusterMass = salmon::math::logAdd(logClusterMass, t.mass(false)); ++clusterSize; }
    for (auto t_id : members) {
      Transcript& t = refs[t_id]; const double logTranscriptMass = t.mass(false);
      if (logTranscriptMass == LOG_0) t.projectedCounts = 0;
      else { const double logClusterFraction = logTranscriptMass - logClusterMass; t.projectedCounts = std::exp(logClusterFraction + logClusterCount);
        requiresProjection |= t.projectedCounts > static_cast<double>(t.totalCounts) or t.projectedCounts < static_cast<double>(t.uniqueCounts); }
    }
    if (clusterSize > 1 and requiresProjection) cptr->projectToPolytope(refs);
  }
  for (uint32_t i = 0; i < M; ++i) projected_out[i] = refs[i].projectedCounts;
  if (num_clusters_out) *num_clusters_out = clusters.size();
  (void)num_mapped;
}
