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

struct Txome { std::vector<std::string> names, seqs; std::vector<uint32_t> gene; std::vector<std::string> gene_exons; /* per gene: exons joined by '|' */ };

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
  o->t.gene_exons.resize(n_genes); for (uint32_t g = 0; g < n_genes; ++g) o->t.gene_exons[g] = std::move(gex[g][0]);
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

// ---- decoy genome (SURVEY.md §8d "G3G", configs[3]) -------------------------------------------------------------------------
// n_chrom chromosomes totalling ~total_nt: random background, `repeat_frac` of it overwritten by copies of 1000 repeat families
// (consensus 300-3000 nt, every copy at 80-95 % identity), then every gene's exons written in order with introns between them
// (lognormal, median 1.5 kb, shrunk when the genes would not fit), gene g on chromosome g % n_chrom.  The exons are written last, so
// every transcript is a spliced copy of its gene's locus.
struct Genome { std::vector<std::string> names, seqs; std::vector<uint32_t> gchrom; std::vector<uint64_t> gstart, gend; };
struct sqs_genome { Genome g; };

sqs_genome* sqs_genome_generate(const sqs_txome* t, uint64_t seed, uint64_t total_nt, uint32_t n_chrom, double repeat_frac, uint32_t nthreads) {
  const Txome& T = t->t; const uint32_t G = (uint32_t)T.gene_exons.size();
  sqs_genome* o = new sqs_genome(); Genome& X = o->g;
  if (n_chrom == 0) n_chrom = 1;
  const uint64_t clen = std::max<uint64_t>(total_nt / n_chrom, 2000);
  X.names.resize(n_chrom); X.seqs.resize(n_chrom); X.gchrom.assign(G, 0); X.gstart.assign(G, 0); X.gend.assign(G, 0);
  // exon lengths per gene and the intron scale that lets the genes take at most 70 % of a chromosome
  std::vector<std::vector<uint32_t>> elen(G); uint64_t exon_nt = 0, nintr = 0;
  for (uint32_t g = 0; g < G; ++g) { const std::string& j = T.gene_exons[g]; size_t p = 0; while (p <= j.size()) { size_t e = j.find('|', p); if (e == std::string::npos) e = j.size(); elen[g].push_back((uint32_t)(e - p)); exon_nt += e - p; p = e + 1; } nintr += elen[g].size() - 1; }
  const double mean_intron = 1500.0 * std::exp(0.5 * 0.8 * 0.8);   // lognormal(median 1500, sigma 0.8)
  double room = 0.7 * (double)clen * n_chrom - (double)exon_nt; if (room < 0) room = 0;
  const double iscale = nintr ? std::min(1.0, room / (mean_intron * (double)nintr)) : 1.0;
  // repeat families (shared by all chromosomes)
  std::vector<std::string> fam(1000);
  { Rng r(seed ^ 0x5EED0FA3ULL); for (auto& f : fam) { uint32_t L = 300 + r.below(2700); f.resize(L); for (auto& c : f) c = ACGT[r.below(4)]; } }
  std::atomic<uint32_t> next(0); std::vector<std::thread> th;
  for (uint32_t tt = 0; tt < std::max(1u, std::min(nthreads, n_chrom)); ++tt) th.emplace_back([&]() {
    for (;;) {
      const uint32_t c = next.fetch_add(1); if (c >= n_chrom) break;
      Rng r(seed ^ (0xC0FFEEULL + (uint64_t)c * 0x9E3779B97F4A7C15ULL));
      std::string& s = X.seqs[c]; s.resize(clen);
      for (uint64_t p = 0; p < clen; p += 32) { uint64_t w = r.next(); const uint64_t e = std::min<uint64_t>(clen, p + 32); for (uint64_t q = p; q < e; ++q, w >>= 2) s[q] = ACGT[w & 3]; }
      // repeats
      uint64_t rep_nt = 0; const uint64_t rep_goal = (uint64_t)(repeat_frac * (double)clen);
      while (rep_nt < rep_goal) {
        const std::string& f = fam[r.below(1000)]; if (f.size() + 1 >= clen) break;
        const uint64_t pos = (uint64_t)(r.u() * (double)(clen - f.size())); const double div = 0.05 + 0.15 * r.u();
        // a copy: substitutions drawn by geometric gaps (no RNG call per base)
        memcpy(&s[pos], f.data(), f.size());
        for (double q = -std::log(1.0 - r.u()) / div; q < (double)f.size(); q += 1.0 - std::log(1.0 - r.u()) / div) s[pos + (uint64_t)q] = ACGT[r.below(4)];
        rep_nt += f.size();
      }
      // genes of this chromosome, in order, evenly spaced
      std::vector<uint32_t> mine; for (uint32_t g = c; g < G; g += n_chrom) mine.push_back(g);
      std::vector<std::vector<uint32_t>> intr(mine.size()); uint64_t span_sum = 0;
      for (size_t i = 0; i < mine.size(); ++i) { const uint32_t g = mine[i]; uint64_t sp = 0; for (size_t e = 0; e < elen[g].size(); ++e) { sp += elen[g][e]; if (e + 1 < elen[g].size()) { double v = 1500.0 * std::exp(0.8 * r.normal()) * iscale; uint32_t il = (uint32_t)std::min(50000.0, std::max(20.0, v)); intr[i].push_back(il); sp += il; } } span_sum += sp; }
      const uint64_t gap = (clen > span_sum) ? (clen - span_sum) / (mine.size() + 1) : 0;
      uint64_t pos = gap;
      for (size_t i = 0; i < mine.size(); ++i) {
        const uint32_t g = mine[i]; const std::string& j = T.gene_exons[g]; size_t p = 0; uint64_t q = pos;
        X.gchrom[g] = c; X.gstart[g] = std::min<uint64_t>(q, clen);
        for (size_t e = 0; e < elen[g].size(); ++e) {
          const uint32_t L = elen[g][e];
          if (q + L <= clen) memcpy(&s[q], j.data() + p, L);
          q += L; p += L + 1; if (e + 1 < elen[g].size()) q += intr[i][e];
        }
        X.gend[g] = std::min<uint64_t>(q, clen); pos = q + gap;
      }
      char nm[32]; snprintf(nm, sizeof(nm), "chr%u", c + 1); X.names[c] = nm;
    }
  });
  for (auto& x : th) x.join();
  return o;
}
void sqs_genome_free(sqs_genome* g) { delete g; }
uint32_t sqs_genome_count(const sqs_genome* g) { return (uint32_t)g->g.names.size(); }
const char* sqs_genome_name(const sqs_genome* g, uint32_t i) { return g->g.names[i].c_str(); }
const char* sqs_genome_seq(const sqs_genome* g, uint32_t i) { return g->g.seqs[i].data(); }
uint64_t sqs_genome_len(const sqs_genome* g, uint32_t i) { return (uint64_t)g->g.seqs[i].size(); }

// sqs_reads_generate with a third source: a fraction `genomic_frac` of the pairs is drawn from the decoy genome, uniformly inside a
// random gene's locus (exons + introns: mostly intronic sequence), truth_tid = 0xFFFFFFFE
void sqs_reads_generate_decoy(const sqs_txome* t, const sqs_genome* gn, uint64_t seed, uint64_t first_pair, uint64_t n_pairs, uint32_t read_len,
                              double sub_rate, double indel_rate, double junk_frac, double genomic_frac, uint8_t* seq, uint32_t* truth_tid,
                              uint32_t* truth_pos, uint32_t nthreads) {
  sqs_reads_generate(t, seed, first_pair, n_pairs, read_len, sub_rate, indel_rate, junk_frac, seq, truth_tid, truth_pos, nthreads);
  if (!gn || genomic_frac <= 0) return;
  const Genome& X = gn->g; const uint32_t G = (uint32_t)X.gstart.size(); if (!G) return;
  std::atomic<uint64_t> next(0); std::vector<std::thread> th;
  for (uint32_t tt = 0; tt < std::max(1u, nthreads); ++tt) th.emplace_back([&]() {
    for (;;) {
      uint64_t b = next.fetch_add(4096); if (b >= n_pairs) break; const uint64_t e = std::min(n_pairs, b + 4096);
      for (uint64_t i = b; i < e; ++i) {
        Rng r(seed ^ 0x6E0D1CULL ^ ((first_pair + i) * 0xD6E8FEB86659FD93ULL));
        if (r.u() >= genomic_frac) continue;
        const uint32_t g = r.below(G); const std::string& s = X.seqs[X.gchrom[g]];
        uint32_t fl = (uint32_t)std::max((double)read_len, 250.0 + 25.0 * r.normal() + 0.5);
        const uint64_t lo = X.gstart[g], hi = X.gend[g]; if (hi < lo + fl + 1) continue;
        const uint64_t st = lo + (uint64_t)(r.u() * (double)(hi - lo - fl));
        uint8_t* o1 = seq + (2 * i) * read_len; uint8_t* o2 = o1 + read_len; const bool flip = r.next() & 1;
        for (uint32_t p = 0; p < read_len; ++p) {
          char a = s[st + p], c2 = comp(s[st + fl - 1 - p]);
          if (r.u() < sub_rate) a = ACGT[r.below(4)]; if (r.u() < sub_rate) c2 = ACGT[r.below(4)];
          (flip ? o2 : o1)[p] = (uint8_t)a; (flip ? o1 : o2)[p] = (uint8_t)c2;
        }
        if (truth_tid) truth_tid[i] = 0xFFFFFFFEu; if (truth_pos) truth_pos[i] = (uint32_t)st;
      }
    }
  });
  for (auto& x : th) x.join();
}
}
