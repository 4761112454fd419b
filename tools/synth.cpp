// tools/synth.cpp — seeded synthetic transcriptome + paired-end read generator.
// There is no GENCODE / real FASTQ offline; this produces a human-transcriptome-*shaped* input:
// genes of 4-25 exons, ~10 isoforms per gene sharing exons (multi-mapping + branching cDBG),
// 2 % paralog families at 90-97 % identity, poly-A tails on 30 % of transcripts; reads with
// log-normal expression, N(250,25) fragments, 0.5 % substitutions, 0.01 % indels, 1 % junk pairs.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL) { next(); next(); }
  inline uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
  inline double u() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  inline uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
  inline double normal() { double a = u(), b = u(); if (a < 1e-300) a = 1e-300; return std::sqrt(-2.0 * std::log(a)) * std::cos(6.283185307179586 * b); }
};
const char ACGT[4] = {'A', 'C', 'G', 'T'};
inline char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

struct Txome { std::vector<std::string> names, seqs; std::vector<uint32_t> gene; };

void gen_gene(uint64_t seed, uint32_t g, uint32_t iso_target, const std::string* paralog_of, std::vector<std::string>& exons_out,
              std::vector<std::string>& names, std::vector<std::string>& seqs) {
  Rng r(seed ^ ((uint64_t)g * 0xD1B54A32D192ED03ULL));
  std::vector<std::string> exons;
  if (paralog_of) {
    // paralog: mutate the source gene's exons at 3-10 % divergence
    double div = 0.03 + 0.07 * r.u();
    size_t p = 0; const std::string& src = *paralog_of;
    while (p < src.size()) { size_t e = src.find('|', p); if (e == std::string::npos) e = src.size(); std::string x = src.substr(p, e - p); for (auto& c : x) if (r.u() < div) c = ACGT[r.below(4)]; exons.push_back(x); p = e + 1; }
  } else {
    uint32_t ne = 4 + r.below(22);
    double gc = 0.42 + 0.13 * r.u();
    for (uint32_t i = 0; i < ne; ++i) {
      double l = 140.0 * std::exp(0.6 * r.normal()); uint32_t len = (uint32_t)std::min(3000.0, std::max(31.0, l));
      std::string x(len, 'A');
      for (auto& c : x) { double v = r.u(); c = (v < gc) ? ((r.next() & 1) ? 'G' : 'C') : ((r.next() & 1) ? 'A' : 'T'); }
      exons.push_back(x);
    }
  }
  { std::string j; for (size_t i = 0; i < exons.size(); ++i) { if (i) j.push_back('|'); j += exons[i]; } exons_out.push_back(j); }
  uint32_t niso = std::max(1u, (uint32_t)(iso_target * (0.5 + r.u())));
  uint32_t constitutive = r.below((uint32_t)exons.size());
  for (uint32_t k = 0; k < niso; ++k) {
    std::string s; double pinc = (k == 0) ? 1.0 : 0.55 + 0.4 * r.u();
    for (uint32_t i = 0; i < exons.size(); ++i) if (i == constitutive || r.u() < pinc) s += exons[i];
    if (r.u() < 0.30) s.append(20 + r.below(180), 'A');
    char nm[64]; snprintf(nm, sizeof(nm), "G%06u.T%02u", g, k);
    names.emplace_back(nm); seqs.push_back(std::move(s));
  }
}
}  // namespace

extern "C" {
struct sqs_txome { Txome t; };

sqs_txome* sqs_txome_generate(uint64_t seed, uint32_t n_genes, uint32_t iso_per_gene, uint32_t nthreads) {
  sqs_txome* o = new sqs_txome();
  std::vector<std::vector<std::string>> gn(n_genes), gs(n_genes), gex(n_genes);
  // paralogs: 2 % of genes copy an earlier *non-paralog* gene; generate sources first
  std::vector<int32_t> src(n_genes, -1);
  { Rng r(seed ^ 0xABCDEF); for (uint32_t g = 1; g < n_genes; ++g) if (r.u() < 0.02) { uint32_t s = r.below(g); if (src[s] < 0) src[g] = (int32_t)s; } }
  auto run = [&](bool paralogs) {
    std::atomic<uint32_t> next(0); std::vector<std::thread> th;
    for (uint32_t t = 0; t < std::max(1u, nthreads); ++t) th.emplace_back([&]() {
      for (;;) { uint32_t g = next.fetch_add(1); if (g >= n_genes) break; if ((src[g] >= 0) != paralogs) continue;
        gen_gene(seed, g, iso_per_gene, paralogs ? &gex[src[g]][0] : nullptr, gex[g], gn[g], gs[g]); } });
    for (auto& x : th) x.join();
  };
  run(false); run(true);
  for (uint32_t g = 0; g < n_genes; ++g) for (size_t i = 0; i < gn[g].size(); ++i) { o->t.names.push_back(std::move(gn[g][i])); o->t.seqs.push_back(std::move(gs[g][i])); o->t.gene.push_back(g); }
  return o;
}
void sqs_txome_free(sqs_txome* t) { delete t; }
uint32_t sqs_txome_count(const sqs_txome* t) { return (uint32_t)t->t.names.size(); }
const char* sqs_txome_name(const sqs_txome* t, uint32_t i) { return t->t.names[i].c_str(); }
const char* sqs_txome_seq(const sqs_txome* t, uint32_t i) { return t->t.seqs[i].data(); }
uint32_t sqs_txome_len(const sqs_txome* t, uint32_t i) { return (uint32_t)t->t.seqs[i].size(); }
uint64_t sqs_txome_total_nt(const sqs_txome* t) { uint64_t s = 0; for (auto& x : t->t.seqs) s += x.size(); return s; }
// fill pointer tables for sq_index_build_mem
void sqs_txome_tables(const sqs_txome* t, const char** names, const char** seqs, uint32_t* lens) {
  for (size_t i = 0; i < t->t.names.size(); ++i) { names[i] = t->t.names[i].c_str(); seqs[i] = t->t.seqs[i].data(); lens[i] = (uint32_t)t->t.seqs[i].size(); }
}
int sqs_txome_write_fasta(const sqs_txome* t, const char* path) {
  FILE* f = fopen(path, "w"); if (!f) return -1;
  for (size_t i = 0; i < t->t.names.size(); ++i) { fprintf(f, ">%s\n", t->t.names[i].c_str()); const std::string& s = t->t.seqs[i]; for (size_t p = 0; p < s.size(); p += 80) { fwrite(s.data() + p, 1, std::min<size_t>(80, s.size() - p), f); fputc('\n', f); } }
  fclose(f); return 0;
}

// Paired reads: seq must hold 2*n*read_len bytes (records are fixed-length read_len; record 2i =
// mate 1, 2i+1 = mate 2); truth_tid[n] (0xFFFFFFFF for junk), truth_pos[n] fragment start.
void sqs_reads_generate(const sqs_txome* t, uint64_t seed, uint64_t first_pair, uint64_t n_pairs, uint32_t read_len,
                        double sub_rate, double indel_rate, double junk_frac, uint8_t* seq, uint32_t* truth_tid, uint32_t* truth_pos, uint32_t nthreads) {
  const Txome& T = t->t; const uint32_t M = (uint32_t)T.seqs.size();
  // expression ~ lognormal(0, 2), 35 % zero; sampling weight = expr * max(len - 250 + 1, 1)
  std::vector<double> cum(M + 1, 0.0);
  { Rng r(seed ^ 0x5151515151ULL); for (uint32_t i = 0; i < M; ++i) { double e = (r.u() < 0.35) ? 0.0 : std::exp(2.0 * r.normal()); double L = (double)T.seqs[i].size(); double w = (L >= read_len + 20) ? e * std::max(L - 250.0 + 1.0, 1.0) : 0.0; cum[i + 1] = cum[i] + w; } }
  const double tot = cum[M];
  std::atomic<uint64_t> next(0); std::vector<std::thread> th;
  for (uint32_t tt = 0; tt < std::max(1u, nthreads); ++tt) th.emplace_back([&]() {
    std::string frag, m1, m2;
    for (;;) {
      uint64_t b = next.fetch_add(4096); if (b >= n_pairs) break; uint64_t e = std::min(n_pairs, b + 4096);
      for (uint64_t i = b; i < e; ++i) {
        Rng r(seed ^ ((first_pair + i) * 0xA24BAED4963EE407ULL));
        uint8_t* o1 = seq + (2 * i) * read_len; uint8_t* o2 = o1 + read_len;
        if (r.u() < junk_frac || tot <= 0) { for (uint32_t p = 0; p < read_len; ++p) { o1[p] = ACGT[r.below(4)]; o2[p] = ACGT[r.below(4)]; } if (truth_tid) truth_tid[i] = 0xFFFFFFFFu; if (truth_pos) truth_pos[i] = 0; continue; }
        double x = r.u() * tot; uint32_t tid = (uint32_t)(std::upper_bound(cum.begin(), cum.end(), x) - cum.begin()) - 1; if (tid >= M) tid = M - 1;
        const std::string& s = T.seqs[tid]; uint32_t L = (uint32_t)s.size();
        uint32_t fl; for (int tries = 0;; ++tries) { double v = 250.0 + 25.0 * r.normal(); fl = (uint32_t)std::max(0.0, v + 0.5); if ((fl >= read_len && fl <= std::min(1000u, L)) || tries > 50) break; }
        if (fl < read_len) fl = read_len; if (fl > L) fl = L;
        uint32_t st = r.below(L - fl + 1);
        frag.assign(s, st, fl);
        auto mutate = [&](const std::string& src, uint8_t* out) {  // src has >= read_len bases available
          uint32_t p = 0, q = 0;
          while (q < read_len) {
            if (p >= src.size()) { out[q++] = ACGT[r.below(4)]; continue; }
            double v = r.u();
            if (v < indel_rate * 0.5) { ++p; continue; }                        // deletion
            if (v < indel_rate) { out[q++] = ACGT[r.below(4)]; continue; }      // insertion
            char c = src[p++]; if (r.u() < sub_rate) { char d; do d = ACGT[r.below(4)]; while (d == c); c = d; }
            out[q++] = (uint8_t)c;
          }
        };
        m1.assign(frag, 0, std::min<uint32_t>(fl, read_len + 8));
        m2.clear(); { uint32_t take = std::min<uint32_t>(fl, read_len + 8); for (uint32_t p = 0; p < take; ++p) m2.push_back(comp(frag[fl - 1 - p])); }
        bool flip = r.next() & 1;  // unstranded: which mate is forward
        mutate(flip ? m2 : m1, o1); mutate(flip ? m1 : m2, o2);
        if (truth_tid) truth_tid[i] = tid; if (truth_pos) truth_pos[i] = st;
      }
    }
  });
  for (auto& x : th) x.join();
}
}
