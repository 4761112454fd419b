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
#include <atomic>
#include <cmath>
#include <cstdio>
#include <map>
#include <numeric>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <algorithm>
#include <memory>
#include <string>
#include <cstring>
#include <cctype>
#include <sys/stat.h>

namespace {
// Lock-free union-find (the larger root is always linked under the smaller one, so every component ends up rooted at
// its smallest transcript id whatever the interleaving): classes are united by several threads at once.
struct DSU {
  std::unique_ptr<std::atomic<uint32_t>[]> p; uint32_t n;
  explicit DSU(uint32_t n_) : p(new std::atomic<uint32_t>[n_]), n(n_) {
    for (uint32_t i = 0; i < n; ++i) p[i].store(i, std::memory_order_relaxed);
  }
  uint32_t root(uint32_t x) {
    for (;;) {
      uint32_t px = p[x].load(std::memory_order_relaxed); if (px == x) return x;
      uint32_t gp = p[px].load(std::memory_order_relaxed);
      if (gp != px) p[x].compare_exchange_weak(px, gp, std::memory_order_relaxed);   // path halving; losing the race is harmless
      x = gp;
    }
  }
  void join(uint32_t a, uint32_t b) {
    for (;;) {
      a = root(a); b = root(b); if (a == b) return;
      if (a > b) std::swap(a, b);
      uint32_t expect = b;
      if (p[b].compare_exchange_strong(expect, a, std::memory_order_relaxed)) return;   // b was still a root: now under a
    }
  }
};
}  // namespace

extern "C" int sq_normalize_alphas(uint32_t M, const sq_eq_table* eq, const double* log_mass, const uint64_t* uniq, const uint64_t* total,
    double* projected) {
  if (!eq || !log_mass || !uniq || !total || !projected) { sq_set_error("sq_normalize_alphas: bad arguments"); return SQ_ERR_ARG; }
  const bool timing = getenv("SQ_TIMING") != nullptr; auto tm0 = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sq-timing] normalize %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tm0).count());
    tm0 = t1;
  };
  const uint32_t nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  DSU d(M); std::atomic<uint32_t> bad{0};
  sq_parallel_for(eq->num_classes, nthr, 8192, [&](uint64_t c_lo, uint64_t c_hi, uint32_t) {
    for (uint64_t c = c_lo; c < c_hi; ++c) {
      const uint64_t a = eq->off[c], b = eq->off[c + 1];
      for (uint64_t i = a; i < b; ++i) if (eq->tid[i] >= M) {
        bad.store(eq->tid[i] + 1 ? eq->tid[i] + 1 : 1, std::memory_order_relaxed);
        goto next;
      }
      for (uint64_t i = a + 1; i < b; ++i) d.join(eq->tid[a], eq->tid[i]);
      next:;
    }
  });
  if (bad.load()) { sq_set_error("sq_normalize_alphas: a label names transcript %u >= %u", bad.load() - 1, M); return SQ_ERR_ARG; }
  std::vector<uint32_t> rootOf(M);
  sq_parallel_for(M, nthr, 8192, [&](uint64_t lo, uint64_t hi,
      uint32_t) { for (uint64_t t = lo; t < hi; ++t) rootOf[t] = d.root((uint32_t)t); });
  mark("union-find");
  // hits per cluster: class counts are integers, so the (atomic, any-order) integer sum is the reference's sum exactly
  std::unique_ptr<std::atomic<uint64_t>[]> hitsI(new std::atomic<uint64_t>[M]);
  for (uint32_t t = 0; t < M; ++t) hitsI[t].store(0, std::memory_order_relaxed);
  sq_parallel_for(eq->num_classes, nthr, 8192, [&](uint64_t c_lo, uint64_t c_hi, uint32_t) {
    for (uint64_t c = c_lo; c < c_hi; ++c) if (eq->off[c + 1] > eq->off[c]) hitsI[rootOf[eq->tid[eq->off[c]]]].fetch_add(eq->count[c],
        std::memory_order_relaxed);
  });
  std::vector<double> hits(M); for (uint32_t t = 0; t < M; ++t) hits[t] = (double)hitsI[t].load(std::memory_order_relaxed);
  // bucket members by root (counting sort keeps ascending tid inside each cluster)
  std::vector<uint32_t> start(M + 1, 0), order(M);
  for (uint32_t t = 0; t < M; ++t) start[rootOf[t] + 1]++;
  for (uint32_t r = 0; r < M; ++r) start[r + 1] += start[r];
  { std::vector<uint32_t> cur(start.begin(), start.end() - 1); for (uint32_t t = 0; t < M; ++t) order[cur[rootOf[t]]++] = t; }
  mark("bucket");
  // clusters are independent (each writes only its members' projected counts; sums inside a cluster keep their order)
  sq_parallel_for(M, nthr, 4096, [&](uint64_t r_lo, uint64_t r_hi, uint32_t) {
  std::vector<uint8_t> bound;
  for (uint32_t r = (uint32_t)r_lo; r < (uint32_t)r_hi; ++r) {
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
  });
  mark("clusters");
  return SQ_OK;
}

// quant.sf: Name Length EffectiveLength TPM NumReads (GZipWriter.cpp:698-736)
extern "C" int sq_write_quant_sf(const char* path, const sq_index* idx, const double* eff_len, const double* num_reads,
    double num_mapped_frags) {
  return sq_write_quant_sf_digits(path, idx, eff_len, num_reads, num_mapped_frags, 3);   // salmon::defaults::sigDigits
}
extern "C" int sq_write_quant_sf_digits(const char* path, const sq_index* idx, const double* eff_len, const double* num_reads,
    double num_mapped_frags, int sig_digits) {
  if (!path || !idx || !eff_len || !num_reads || sig_digits < 0 || sig_digits > 30) { sq_set_error("sq_write_quant_sf: bad arguments"); return SQ_ERR_ARG; }
  FILE* f = fopen(path, "w"); if (!f) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  // decoys are dropped before inference and output (readExp.dropDecoyTranscripts(), SalmonQuantify.cpp:2479): targets only
  const uint32_t M = std::min<uint32_t>((uint32_t)idx->names.size(), idx->first_decoy);
  // explicitSum (:704-708)
  if (!(num_mapped_frags > 0)) {
    num_mapped_frags = 0;
    for (uint32_t i = 0; i < M; ++i) num_mapped_frags += num_reads[i];
  }
  double denom = 0.0;
  for (uint32_t i = 0; i < M; ++i) denom += (num_reads[i] / num_mapped_frags) / eff_len[i];                                       // :718-722
  fprintf(f, "Name\tLength\tEffectiveLength\tTPM\tNumReads\n");
  for (uint32_t i = 0; i < M; ++i) {
    double npm = num_reads[i] / num_mapped_frags; double tpm = denom > 0 ? ((npm / eff_len[i]) / denom) * 1000000.0 : 0.0;
    fprintf(f, "%s\t%u\t%.*f\t%f\t%.*f\n", idx->names[i].c_str(), idx->ref_clen[i], sig_digits, eff_len[i], tpm, sig_digits, num_reads[i]);   // --sigDigits (GZipWriter.cpp:734-736)
  }
  fclose(f);
  return SQ_OK;
}

// aux_info/eq_classes.txt.gz (doc/source/file_formats.rst:173-253). with_weights = --dumpEqWeights;
// otherwise range-factorised classes are collapsed to transcript sets (GZipWriter.cpp:89-114).
extern "C" int sq_write_eq_classes(const char* path, const sq_index* idx, const sq_eq_table* eq, int with_weights) {
  if (!path || !idx || !eq) { sq_set_error("sq_write_eq_classes: bad arguments"); return SQ_ERR_ARG; }
  gzFile g = gzopen(path, "wb"); if (!g) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  // decoys are dropped before inference and output (readExp.dropDecoyTranscripts(), SalmonQuantify.cpp:2479): targets only
  const uint32_t M = std::min<uint32_t>((uint32_t)idx->names.size(), idx->first_decoy);
  if (with_weights) {
    gzprintf(g, "%u\n%llu\n", M, (unsigned long long)eq->num_classes);
    for (uint32_t t = 0; t < M; ++t) gzprintf(g, "%s\n", idx->names[t].c_str());
    for (uint64_t c = 0; c < eq->num_classes; ++c) {
      const uint64_t a = eq->off[c], b = eq->off[c + 1];
      gzprintf(g, "%llu\t", (unsigned long long)(b - a));
      for (uint64_t i = a; i < b; ++i) gzprintf(g, "%u\t", eq->tid[i]);
      for (uint64_t i = a; i < b; ++i) gzprintf(g, "%.17g\t", eq->w[i]);
      gzprintf(g, "%llu\n", (unsigned long long)eq->count[c]);
    }
  } else {
    std::map<std::vector<uint32_t>, uint64_t> col;
    for (uint64_t c = 0; c < eq->num_classes; ++c) {
      std::vector<uint32_t> k(eq->tid + eq->off[c], eq->tid + eq->off[c + 1]);
      col[k] += eq->count[c];
    }
    gzprintf(g, "%u\n%zu\n", M, col.size());
    for (uint32_t t = 0; t < M; ++t) gzprintf(g, "%s\n", idx->names[t].c_str());
    for (auto& kv : col) {
      gzprintf(g, "%zu\t", kv.first.size());
      for (uint32_t t : kv.first) gzprintf(g, "%u\t", t);
      gzprintf(g, "%llu\n", (unsigned long long)kv.second);
    }
  }
  gzclose(g);
  return SQ_OK;
}

// ---- `salmon quant -e` input: aux_info/eq_classes.txt[.gz] written with --dumpEqWeights -------------
// salmon::utils::readEquivCounts (reference src/util/SalmonUtils.cpp:1026-1122): M, E, M names, E rows
// "n  tid*n  weight*n  count", then optional "name effLen" pairs; missing effective lengths are 100.
struct sq_eq_file { std::vector<std::string> names; std::vector<double> eff; std::vector<uint64_t> off,
    count; std::vector<uint32_t> tid; std::vector<double> w; };
extern "C" int sq_eq_file_read(const char* path, sq_eq_file** out) {
  if (!path || !out) { sq_set_error("sq_eq_file_read: bad arguments"); return SQ_ERR_ARG; }
  gzFile g = gzopen(path, "rb"); if (!g) { sq_set_error("cannot read '%s'", path); return SQ_ERR_IO; }   // gzopen reads plain text too
  std::string all; char buf[1 << 16]; int n;
  while ((n = gzread(g, buf, sizeof(buf))) > 0) all.append(buf, (size_t)n);
  gzclose(g);
  const char* p = all.c_str(); const char* end = p + all.size();
  auto tok = [&](std::string& t) {
    while (p < end && isspace((unsigned char)*p)) ++p;
    const char* b = p;
    while (p < end && !isspace((unsigned char)*p)) ++p;
    t.assign(b, p);
    return !t.empty();
  };
  std::string t; std::unique_ptr<sq_eq_file> F(new sq_eq_file());
  if (!tok(t)) { sq_set_error("'%s': empty eq-class file", path); return SQ_ERR_IO; } const uint64_t M = strtoull(t.c_str(), nullptr, 10);
  if (!tok(t)) { sq_set_error("'%s': truncated header", path); return SQ_ERR_IO; } const uint64_t E = strtoull(t.c_str(), nullptr, 10);
  std::map<std::string, size_t> idx;
  for (uint64_t i = 0; i < M; ++i) {
    if (!tok(t)) {
      sq_set_error("'%s': truncated name list", path);
      return SQ_ERR_IO;
    }
    idx[t] = F->names.size();
    F->names.push_back(t);
  }
  F->off.assign(1, 0);
  for (uint64_t c = 0; c < E; ++c) {
    if (!tok(t)) { sq_set_error("'%s': truncated at class %llu", path, (unsigned long long)c); return SQ_ERR_IO; }
    const uint64_t k = strtoull(t.c_str(), nullptr, 10);
    for (uint64_t i = 0; i < k; ++i) {
      if (!tok(t)) {
        sq_set_error("'%s': truncated labels", path);
        return SQ_ERR_IO;
      }
      const uint64_t x = strtoull(t.c_str(), nullptr, 10);
      if (x >= M) {
        sq_set_error("'%s': transcript id %llu out of range", path, (unsigned long long)x);
        return SQ_ERR_IO;
      }
      F->tid.push_back((uint32_t)x);
    }
    for (uint64_t i = 0; i < k; ++i) {
      if (!tok(t)) {
        sq_set_error("'%s': class %llu has no weights (write the file with --dumpEqWeights)", path, (unsigned long long)c);
        return SQ_ERR_IO;
      }
      F->w.push_back(strtod(t.c_str(), nullptr));
    }
    if (!tok(t)) { sq_set_error("'%s': class %llu has no count", path, (unsigned long long)c); return SQ_ERR_IO; }
    F->count.push_back(strtoull(t.c_str(), nullptr, 10)); F->off.push_back(F->tid.size());
  }
  F->eff.assign(M, 100.0);
  std::string nm;
  while (tok(nm)) {
    if (!tok(t)) break;
    auto it = idx.find(nm);
    if (it == idx.end()) {
      sq_set_error("'%s': effective length for unknown transcript '%s'", path, nm.c_str());
      return SQ_ERR_IO;
    }
    F->eff[it->second] = strtod(t.c_str(), nullptr);
  }
  *out = F.release();
  return SQ_OK;
}
extern "C" void sq_eq_file_free(sq_eq_file* f) { delete f; }
extern "C" uint32_t sq_eq_file_num_txp(const sq_eq_file* f) { return f ? (uint32_t)f->names.size() : 0; }
extern "C" const char* sq_eq_file_name(const sq_eq_file* f, uint32_t i) {
  return (f && i < f->names.size()) ? f->names[i].c_str() : nullptr;
}
extern "C" const double* sq_eq_file_eff_lens(const sq_eq_file* f) { return f ? f->eff.data() : nullptr; }
extern "C" int sq_eq_file_table(const sq_eq_file* f, sq_eq_table* t) {
  if (!f || !t) return SQ_ERR_ARG;
  memset(t, 0, sizeof(*t)); t->num_classes = f->count.size(); t->num_labels = f->tid.size();
  t->off = const_cast<uint64_t*>(f->off.data());
  t->tid = const_cast<uint32_t*>(f->tid.data());
  t->w = const_cast<double*>(f->w.data());
  t->count = const_cast<uint64_t*>(f->count.data());
  return SQ_OK;
}

// quant.sf from plain name / length arrays (the -e mode has no index)
extern "C" int sq_write_quant_sf_names(const char* path, uint32_t M, const char* const* names, const uint32_t* lens, const double* eff_len,
    const double* num_reads,
    double num_mapped_frags) {
  if (!path || !names || !eff_len || !num_reads) { sq_set_error("sq_write_quant_sf_names: bad arguments"); return SQ_ERR_ARG; }
  FILE* f = fopen(path, "w"); if (!f) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  if (!(num_mapped_frags > 0)) { num_mapped_frags = 0; for (uint32_t i = 0; i < M; ++i) num_mapped_frags += num_reads[i]; }
  double denom = 0.0;
  for (uint32_t i = 0; i < M; ++i) denom += (num_reads[i] / num_mapped_frags) / eff_len[i];
  fprintf(f, "Name\tLength\tEffectiveLength\tTPM\tNumReads\n");
  for (uint32_t i = 0; i < M; ++i) {
    double npm = num_reads[i] / num_mapped_frags; double tpm = denom > 0 ? ((npm / eff_len[i]) / denom) * 1000000.0 : 0.0;
    fprintf(f, "%s\t%u\t%.3f\t%f\t%.3f\n", names[i], lens ? lens[i] : (uint32_t)eff_len[i], eff_len[i], tpm, num_reads[i]);
  }
  fclose(f);
  return SQ_OK;
}

// aux_info/bootstrap/{names.tsv.gz, bootstraps.gz}: one tab-separated name row; raw f64[M] per replicate,
// appended in arrival order (GZipWriter.cpp:306-326, 765-788)
struct sq_boot_writer { gzFile g = nullptr; uint32_t M = 0; uint64_t written = 0; };
extern "C" int sq_boot_writer_open(const char* aux_dir, uint32_t M, const char* const* names, sq_boot_writer** out) {
  if (!aux_dir || !names || !out || M == 0) { sq_set_error("sq_boot_writer_open: bad arguments"); return SQ_ERR_ARG; }
  const std::string d = std::string(aux_dir) + "/bootstrap";
  mkdir(aux_dir, 0755); mkdir(d.c_str(), 0755);
  gzFile n = gzopen((d + "/names.tsv.gz").c_str(), "wb6");
  if (!n) {
    sq_set_error("cannot write '%s/names.tsv.gz'", d.c_str());
    return SQ_ERR_IO;
  }
  for (uint32_t i = 0; i < M; ++i) { gzputs(n, names[i]); if (i + 1 < M) gzputc(n, '\t'); }
  gzputc(n, '\n'); gzclose(n);
  std::unique_ptr<sq_boot_writer> W(new sq_boot_writer()); W->M = M;
  W->g = gzopen((d + "/bootstraps.gz").c_str(), "wb6");
  if (!W->g) {
    sq_set_error("cannot write '%s/bootstraps.gz'", d.c_str());
    return SQ_ERR_IO;
  }
  *out = W.release();
  return SQ_OK;
}
extern "C" int sq_boot_writer_append(sq_boot_writer* w, const double* alphas, uint32_t M) {
  if (!w || !w->g || !alphas || M != w->M) { sq_set_error("sq_boot_writer_append: bad arguments"); return SQ_ERR_ARG; }
  if (gzwrite(w->g, alphas, (unsigned)((size_t)M * 8)) != (int)((size_t)M * 8)) {
    sq_set_error("short write to bootstraps.gz");
    return SQ_ERR_IO;
  }
  w->written++;
  return SQ_OK;
}
extern "C" uint64_t sq_boot_writer_close(sq_boot_writer* w) {
  if (!w) return 0;
  uint64_t n = w->written;
  if (w->g) gzclose(w->g);
  delete w;
  return n;
}

// aux_info/ambig_info.tsv (GZipWriter.cpp:601-638): per transcript, fragments in single-transcript classes and the
// count mass of the multi-transcript classes it belongs to (uint32 accumulators, as in the reference)
extern "C" int sq_write_ambig_info(const char* path, uint32_t M, const sq_eq_table* eq) {
  if (!path || !eq || (eq->num_classes && (!eq->off || !eq->tid || !eq->count))) {
    sq_set_error("sq_write_ambig_info: bad arguments");
    return SQ_ERR_ARG;
  }
  std::vector<uint32_t> uniq(M, 0), amb(M, 0);
  for (uint64_t c = 0; c < eq->num_classes; ++c) {
    const uint64_t a = eq->off[c], b = eq->off[c + 1];
    if (b - a > 1) { for (uint64_t i = a; i < b; ++i) if (eq->tid[i] < M) amb[eq->tid[i]] += (uint32_t)eq->count[c]; }
    else if (b - a == 1 && eq->tid[a] < M) uniq[eq->tid[a]] += (uint32_t)eq->count[c];
  }
  FILE* f = fopen(path, "w"); if (!f) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  fprintf(f, "UniqueCount\tAmbigCount\n");
  for (uint32_t i = 0; i < M; ++i) fprintf(f, "%u\t%u\n", uniq[i], amb[i]);
  fclose(f);
  return SQ_OK;
}

// lib_format_counts.json (ReadExperiment::summarizeLibraryTypeCounts, reference
// include/salmon/internal/quant/ReadExperiment.inl:219-348).  Format id = type | orientation << 1 | strandedness << 3
// with type {0 single, 1 paired}, orientation {0 same, 1 away, 2 toward, 3 none}, strandedness {0 SA, 1 AS, 2 S, 3 A, 4 U}.
static std::string lib_format_name(uint32_t id) {   // the salmon library-type string, empty for combinations LibraryFormat::check() rejects
  const uint32_t type = id & 1, orient = (id >> 1) & 3, strand = id >> 3;
  if (strand > 4) return "";
  if (type == 0) { if (orient != 3) return ""; return strand == 2 ? "SF" : strand == 3 ? "SR" : strand == 4 ? "U" : ""; }
  if (orient == 3) return "";
  const char* o = orient == 0 ? "M" : orient == 1 ? "O" : "I";
  if (orient == 0) {
    if (strand == 2) return std::string(o) + "SF";
    if (strand == 3) return std::string(o) + "SR";
    if (strand == 4) return std::string(o) + "U";
    return "";
  }
  if (strand == 0) return std::string(o) + "SF";
  if (strand == 1) return std::string(o) + "SR";
  if (strand == 4) return std::string(o) + "U";
  return "";
}
extern "C" int sq_write_lib_format_counts(const char* path, const char* read_files, uint8_t lib_type, uint8_t lib_orientation,
    uint8_t lib_strand,
                                          const uint64_t* counts, uint64_t num_assigned, uint64_t num_compatible) {
  if (!path || !counts) { sq_set_error("sq_write_lib_format_counts: bad arguments"); return SQ_ERR_ARG; }
  const uint32_t fid = (uint32_t)lib_type | ((uint32_t)lib_orientation << 1) | ((uint32_t)lib_strand << 3);
  // the two stranded variants of the expected orientation (:247-262)
  const uint32_t s1 = (lib_orientation == 0 || lib_orientation == 3) ? 2u : 0u, s2 = (lib_orientation == 0 ||
      lib_orientation == 3) ? 3u : 1u;
  const uint32_t f1 = (uint32_t)lib_type | ((uint32_t)lib_orientation << 1) | (s1 << 3),
      f2 = (uint32_t)lib_type | ((uint32_t)lib_orientation << 1) | (s2 << 3);
  uint64_t nAgree = 0, nDisStranded = 0, nF1 = 0, nF2 = 0, nDisUnstranded = 0;
  for (uint32_t i = 0; i < 64; ++i) {
    if (i == fid) nAgree = counts[i]; else nDisStranded += counts[i];
    if (i == f1) nF1 = counts[i]; else if (i == f2) nF2 = counts[i]; else nDisUnstranded += counts[i];
  }
  uint64_t nDisagree; double ratio;
  if (lib_strand == 4) { nAgree = nF1 + nF2; nDisagree = nDisUnstranded; ratio = nAgree > 0 ? (double)nF1 / (double)(nF1 + nF2) : 0.0; }
  else { nDisagree = nDisStranded; ratio = nAgree > 0 ? (double)nF1 / (double)(nF1 + nF2) : 0.0; }
  FILE* f = fopen(path, "w"); if (!f) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  fprintf(f,
      "{\n    \"read_files\": \"%s\",\n    \"expected_format\": \"%s\",\n    \"compatible_fragment_ratio\": %.17g,\n    \"num_compatible_fragments\": %llu,\n    \"num_assigned_fragments\": %llu,\n"
             "    \"num_frags_with_concordant_consistent_mappings\": %llu,\n    \"num_frags_with_inconsistent_or_orphan_mappings\": %llu,\n    \"strand_mapping_bias\": %.17g",
          read_files ? read_files : "", lib_format_name(fid).c_str(), num_assigned ? (double)num_compatible / (double)num_assigned : 0.0,
          (unsigned long long)num_compatible, (unsigned long long)num_assigned, (unsigned long long)nAgree, (unsigned long long)nDisagree,
              ratio);
  for (uint32_t i = 0; i < 64; ++i) {
    const std::string d = lib_format_name(i);
    if (!d.empty()) fprintf(f, ",\n    \"%s\": %llu", d.c_str(), (unsigned long long)counts[i]);
  }
  fprintf(f, "\n}\n");
  fclose(f);
  return SQ_OK;
}

// ------------------------------------------------------------------------------------------------
// [r4] the rest of aux_info (GZipWriter::writeMeta, src/output/GZipWriter.cpp:294-599): fld.gz, the legacy k-mer bias vectors, the
// binary bias-model dumps and meta_info.json with the reference's key set in the reference's order.
namespace {
// sq_rng.h's sq_r64 / sq_u01 (that header pulls device intrinsics in; this file is host code)
inline uint64_t fin_r64(uint64_t seed, uint64_t a, uint64_t b) { return sq_mix64(seed ^ sq_mix64(a * 0x9E3779B97F4A7C15ULL + 0x1234567ULL) ^ sq_mix64(b * 0xD1B54A32D192ED03ULL + 0x89ABCDEFULL)); }
inline double fin_u01(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
bool gz_write_all(const std::string& path, const void* p, size_t bytes) {   // writeVectorToFile (:41-56): gzip level 6 of the raw bytes
  gzFile g = gzopen(path.c_str(), "wb6"); if (!g) return false;
  const char* c = (const char*)p; size_t left = bytes; bool ok = true;
  while (left && ok) { const unsigned n = (unsigned)std::min<size_t>(left, 1u << 30); ok = gzwrite(g, c, n) == (int)n; c += n; left -= n; }
  return gzclose(g) == Z_OK && ok;
}
std::string json_escape(const char* s) {
  std::string o; for (; s && *s; ++s) { const unsigned char c = (unsigned char)*s;
    if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); } else if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); o += b; } else o.push_back((char)c); }
  return o;
}
}  // namespace

// distribution_utils::samplesFromLogPMF (src/util/DistributionUtils.cpp:57-102) + writeVectorToFile: the log PMF over [minVal, maxVal] is
// renormalised, `mean` = exp(logsum_i log(i) + logPMF_i) and `sd` from sum p_i i^2 over i in [minVal, maxVal) — the loop's own bounds —,
// then 10 000 draws from the discrete distribution are histogrammed into int32[maxVal + 1] and gzipped.  The reference seeds a Mersenne
// twister from the random device; here draw j is the inverse-CDF of u01(seed, 0xF1D, j) (sq_rng.h): the same file for the same run.
extern "C" int sq_write_fld_samples(const char* path, const double* log_pmf, uint32_t min_len, uint32_t max_len, uint32_t num_samples, uint64_t seed,
                                    double* mean_out, double* sd_out, uint32_t* support_out) {
  if (!log_pmf || max_len < 1 || max_len > 1000000 || min_len > max_len) { sq_set_error("sq_write_fld_samples: bad arguments"); return SQ_ERR_ARG; }
  const size_t minV = min_len, maxV = max_len, n = maxV - minV + 1;
  std::vector<double> lp(log_pmf + minV, log_pmf + maxV + 1);
  double sum = SQ_LOG_0; for (double v : lp) sum = sq_log_add(sum, v);
  for (double& v : lp) v -= sum;
  double mean = SQ_LOG_0, var = 0.0; std::vector<double> pmf(maxV + 1, 0.0);
  for (size_t i = minV; i < maxV; ++i) {
    if (i > 0) mean = sq_log_add(mean, sq_log((double)i) + lp[i - minV]);
    pmf[i] = sq_exp(lp[i - minV]); var += pmf[i] * (double)(i * i);
  }
  mean = sq_exp(mean); var -= mean * mean; const double sd = std::sqrt(var);
  (void)n;
  std::vector<double> cdf(maxV + 1, 0.0); double acc = 0.0; for (size_t i = 0; i <= maxV; ++i) { acc += pmf[i]; cdf[i] = acc; }
  std::vector<int32_t> samples(maxV + 1, 0);
  if (acc > 0.0) for (uint32_t j = 0; j < num_samples; ++j) {
    const double u = fin_u01(fin_r64(seed, 0xF1DULL, j)) * acc;
    size_t k = (size_t)(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin()); if (k > maxV) k = maxV;
    ++samples[k];
  }
  if (mean_out) *mean_out = mean; if (sd_out) *sd_out = sd; if (support_out) *support_out = (uint32_t)samples.size();
  if (path && !gz_write_all(path, samples.data(), samples.size() * sizeof(int32_t))) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  return SQ_OK;
}

// observed_bias.gz / observed_bias_3p.gz / expected_bias.gz (GZipWriter.cpp:335-351): the 6-mer read-start tables of the old bias model.
// Nothing in the reference updates them any more (SalmonQuantify.cpp:1097-1100 are comments), so they hold their initial values: 4^6
// pseudo-counts of 1 (ReadKmerDist.hpp:20-26) and 4^6 expected weights of 1.0 (BiasLibraryState.hpp:35).  num_bias_bins = 4096.
extern "C" int sq_write_legacy_bias(const char* aux_dir, uint32_t* num_bias_bins) {
  if (!aux_dir) { sq_set_error("sq_write_legacy_bias: bad arguments"); return SQ_ERR_ARG; }
  const std::vector<int32_t> obs(4096, 1); const std::vector<double> expd(4096, 1.0); const std::string d(aux_dir);
  if (!gz_write_all(d + "/expected_bias.gz", expd.data(), expd.size() * 8) || !gz_write_all(d + "/observed_bias.gz", obs.data(), obs.size() * 4) ||
      !gz_write_all(d + "/observed_bias_3p.gz", obs.data(), obs.size() * 4)) { sq_set_error("cannot write the bias vectors into '%s'", aux_dir); return SQ_ERR_IO; }
  if (num_bias_bins) *num_bias_bins = 4096;
  return SQ_OK;
}

// GCFragModel::writeBinary (include/salmon/internal/model/GCFragModel.hpp:63-79): int32 dtype (0 linear, 1 log), Eigen::Index rows, cols
// (8 bytes each), `rows` model totals, then the rows x cols counts in Eigen's default column-major order.  `counts` arrives row-major
// [rows][cols] (context class x GC bin), as sq_model_fetch_gc_observed / the bias report hold it.
extern "C" int sq_write_gc_model(const char* path, int32_t dtype, uint32_t rows, uint32_t cols, const double* totals, const double* counts) {
  if (!path || !totals || !counts || !rows || !cols) { sq_set_error("sq_write_gc_model: bad arguments"); return SQ_ERR_ARG; }
  std::vector<char> b; auto put = [&](const void* p, size_t n) { b.insert(b.end(), (const char*)p, (const char*)p + n); };
  const int64_t r = rows, c = cols; put(&dtype, 4); put(&r, 8); put(&c, 8); put(totals, (size_t)rows * 8);
  for (uint32_t j = 0; j < cols; ++j) for (uint32_t i = 0; i < rows; ++i) put(&counts[(size_t)i * cols + j], 8);
  if (!gz_write_all(path, b.data(), b.size())) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  return SQ_OK;
}

// SBModel::writeBinary (src/model/SBModel.cpp:77-116): context length 9 = 3 bases left + the read start + 5 right (SBModel.cpp:21-33), the
// per-position orders {0,1,2,2,2,2,2,2,2}, shifts 18 - 2(i+1) and widths 2(order+1) (:56-59), then the 64 x 9 matrix of log transition
// probabilities (column = position, Eigen's column-major = `log_probs` as sq_bias_eff_lengths returns a model: [9][64]) and the 4 x 9 marginals.
// The marginals are what SBModel::normalize leaves (:233-246): prior 1e-10 plus the mean over a position's 4^order states of the state's
// four transition probabilities, here recovered from the log probabilities (a state that was never seen carries log(1e-5) in the table and
// contributes that, not 0: a dump for inspection, not an input of anything).
extern "C" int sq_write_seq_model(const char* path, const double* lp) {
  if (!path || !lp) { sq_set_error("sq_write_seq_model: bad arguments"); return SQ_ERR_ARG; }
  std::vector<char> b; auto put = [&](const void* p, size_t n) { b.insert(b.end(), (const char*)p, (const char*)p + n); };
  const int32_t len = 9, left = 3, right = 5, order[9] = {0, 1, 2, 2, 2, 2, 2, 2, 2}; int32_t shifts[9], widths[9];
  for (int i = 0; i < 9; ++i) { shifts[i] = 2 * len - 2 * (i + 1); widths[i] = 2 * (order[i] + 1); }
  put(&len, 4); put(&left, 4); put(&right, 4); put(order, 36); put(shifts, 36); put(widths, 36);
  double marg[9][4];
  for (int pos = 0; pos < 9; ++pos) {
    const int states = 1 << (2 * order[pos]);
    for (int x = 0; x < 4; ++x) { double m = 1e-10; for (int st = 0; st < states; ++st) m += std::exp(lp[pos * 64 + st * 4 + x]); marg[pos][x] = m / states; }
  }
  const int64_t pr = 64, pc = 9, mr = 4, mc = 9; put(&pr, 8); put(&pc, 8); put(lp, 9 * 64 * 8); put(&mr, 8); put(&mc, 8); put(marg, sizeof(marg));
  if (!gz_write_all(path, b.data(), b.size())) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  return SQ_OK;
}

// the lambda of GZipWriter.cpp:424-447: uint32 number of models, the length-class bounds, then per model SimplePosBias::writeBinary
// (src/model/SimplePosBias.cpp:84-101): uint32 model length + that many doubles (the finalized masses)
extern "C" int sq_write_pos_models(const char* path, uint32_t num_models, const uint32_t* len_bounds, uint32_t model_len, const double* masses) {
  if (!path || !len_bounds || !masses || !num_models || !model_len) { sq_set_error("sq_write_pos_models: bad arguments"); return SQ_ERR_ARG; }
  std::vector<char> b; auto put = [&](const void* p, size_t n) { b.insert(b.end(), (const char*)p, (const char*)p + n); };
  put(&num_models, 4); put(len_bounds, (size_t)num_models * 4);
  for (uint32_t m = 0; m < num_models; ++m) { put(&model_len, 4); put(masses + (size_t)m * model_len, (size_t)model_len * 8); }
  if (!gz_write_all(path, b.data(), b.size())) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  return SQ_OK;
}

// aux_info/meta_info.json — GZipWriter::writeMeta's keys (GZipWriter.cpp:497-597), same names, same order, same JSON types (cereal's
// layout: four-space indent).  `m->quant_errors` non-NULL = the writeEmptyMeta form's error list (:180-290).
extern "C" int sq_write_meta_info(const char* path, const sq_meta_info* m) {
  if (!path || !m) { sq_set_error("sq_write_meta_info: bad arguments"); return SQ_ERR_ARG; }
  FILE* f = fopen(path, "w"); if (!f) { sq_set_error("cannot write '%s'", path); return SQ_ERR_IO; }
  auto S = [&](const char* s) { return "\"" + json_escape(s ? s : "") + "\""; };
  auto B = [](int v) { return v ? "true" : "false"; };
  fprintf(f, "{\n    \"salmon_version\": %s,\n    \"samp_type\": %s,\n    \"opt_type\": %s,\n    \"quant_errors\": [", S(m->salmon_version ? m->salmon_version : "1.11.4").c_str(), S(m->samp_type).c_str(), S(m->opt_type).c_str());
  if (m->quant_errors && m->quant_errors[0]) fprintf(f, "\n        %s\n    ", S(m->quant_errors).c_str());
  fprintf(f, "],\n    \"num_libraries\": %u,\n    \"library_types\": [", m->num_libraries);
  for (uint32_t i = 0; i < m->num_libraries; ++i) fprintf(f, "%s\n        %s", i ? "," : "", S(m->library_types ? m->library_types[i] : "").c_str());
  fprintf(f, "%s],\n", m->num_libraries ? "\n    " : "");
  fprintf(f, "    \"frag_dist_length\": %u,\n    \"frag_length_mean\": %.17g,\n    \"frag_length_sd\": %.17g,\n    \"seq_bias_correct\": %s,\n    \"gc_bias_correct\": %s,\n    \"num_bias_bins\": %u,\n",
          m->frag_dist_length, m->frag_length_mean, m->frag_length_sd, B(m->seq_bias_correct), B(m->gc_bias_correct), m->num_bias_bins);
  fprintf(f, "    \"mapping_type\": %s,\n", S(m->mapping_type).c_str());
  if (m->keep_duplicates >= 0) fprintf(f, "    \"keep_duplicates\": %s,\n", B(m->keep_duplicates));   // the UNKNOWN status writes no key (:518-529)
  fprintf(f, "    \"num_valid_targets\": %llu,\n    \"num_decoy_targets\": %llu,\n    \"num_eq_classes\": %llu,\n    \"serialized_eq_classes\": %s,\n    \"eq_class_properties\": [",
          (unsigned long long)m->num_valid_targets, (unsigned long long)m->num_decoy_targets, (unsigned long long)m->num_eq_classes, B(m->serialized_eq_classes));
  { std::vector<const char*> props; if (m->range_factorized) props.push_back("range_factorized"); if (m->scalar_weights) props.push_back("scalar_weights"); props.push_back("gzipped");
    for (size_t i = 0; i < props.size(); ++i) fprintf(f, "%s\n        \"%s\"", i ? "," : "", props[i]);
    fprintf(f, "\n    ],\n"); }
  fprintf(f, "    \"length_classes\": [");
  for (uint32_t i = 0; i < m->num_length_classes; ++i) fprintf(f, "%s\n        %u", i ? "," : "", m->length_classes[i]);
  fprintf(f, "%s],\n", m->num_length_classes ? "\n    " : "");
  fprintf(f, "    \"index_seq_hash\": %s,\n    \"index_name_hash\": %s,\n    \"index_seq_hash512\": %s,\n    \"index_name_hash512\": %s,\n    \"index_decoy_seq_hash\": %s,\n    \"index_decoy_name_hash\": %s,\n",
          S(m->index_seq_hash).c_str(), S(m->index_name_hash).c_str(), S(m->index_seq_hash512).c_str(), S(m->index_name_hash512).c_str(), S(m->index_decoy_seq_hash).c_str(), S(m->index_decoy_name_hash).c_str());
  fprintf(f, "    \"num_bootstraps\": %llu,\n    \"num_processed\": %llu,\n    \"num_mapped\": %llu,\n    \"num_decoy_fragments\": %llu,\n    \"num_dovetail_fragments\": %llu,\n"
             "    \"num_fragments_filtered_vm\": %llu,\n    \"num_alignments_below_threshold_for_mapped_fragments_vm\": %llu,\n    \"percent_mapped\": %.17g,\n    \"call\": \"quant\",\n"
             "    \"start_time\": %s,\n    \"end_time\": %s",
          (unsigned long long)m->num_bootstraps, (unsigned long long)m->num_processed, (unsigned long long)m->num_mapped, (unsigned long long)m->num_decoy_fragments,
          (unsigned long long)m->num_dovetail_fragments, (unsigned long long)m->num_fragments_filtered_vm, (unsigned long long)m->num_alignments_below_threshold_vm,
          m->percent_mapped, S(m->start_time).c_str(), S(m->end_time).c_str());
  // keys the reference does not have come last, under one object, so a reader of the reference's keys never meets them in between
  if (m->backend) fprintf(f, ",\n    \"salmon_hip\": {\n        \"backend\": %s,\n        \"num_em_iterations\": %u,\n        \"num_degenerate_eq_classes\": %u,\n        \"pos_bias_correct\": %s,\n        \"runtime_s\": %.3f\n    }",
                          S(m->backend).c_str(), m->num_em_iterations, m->num_degenerate_eq_classes, B(m->pos_bias_correct), m->runtime_s);
  fprintf(f, "\n}\n");
  fclose(f);
  return SQ_OK;
}
