// tools/host_sanitize.cpp — drives the HOST side of the library (index builder + dictionary, read pipeline,
// normalizeAlphas with its lock-free union-find, the file writers / readers) so that it can be run under
// AddressSanitizer + UBSan and under ThreadSanitizer:  make -C tools sanitize   (g++ only, no GPU).
// the device-side FASTQ splitter lives in a .hip file: this host-only build has none and says "no device" (reader.cpp then keeps its own path)
#include "../salmon_amd/csrc/host/reader_dev.h"
int sq_dev_reader_open(const std::vector<std::string>&, const std::vector<std::string>&, uint32_t, uint32_t, sq_dev_reader**) { return SQ_ERR_DEVICE; }
int sq_dev_reader_next(sq_dev_reader*, sq_read_batch*, int*) { return SQ_ERR_DEVICE; }
void sq_dev_reader_release(sq_dev_reader*, int) {}
uint64_t sq_dev_reader_total(const sq_dev_reader*) { return 0; }
void sq_dev_reader_close(sq_dev_reader*) {}
#include "../include/salmon_hip.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <unordered_map>
#include <sys/stat.h>

#include "../salmon_amd/csrc/host/index.h"
// the HIP side (hip/index_dev.hip) is not linked here: host-only stand-ins for its three entry points
void sq_device_index_free(sq_device_index*) {}
int sq_index_breaks_dev(int, const uint64_t*, uint64_t, const uint32_t*, const uint64_t*, uint32_t, uint32_t, uint64_t, uint64_t*, uint64_t*) { return SQ_ERR_DEVICE; }   // [r6] hip/index_build_dev.hip: without it the builder's host passes run
extern "C" int sq_index_load(const char* dir, int, sq_index** out) { return sq_index_load_host(dir, out); }
extern "C" void sq_index_free(sq_index* idx) { delete idx; }
#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED %s:%d: %s  [%s]\n", __FILE__, __LINE__, #x, sq_last_error()); exit(1); } } while (0)

static std::string rnd_seq(std::mt19937_64& g, size_t n) { std::string s(n, 'A'); for (auto& c : s) c = "ACGT"[g() & 3]; return s; }

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp/sq_host_sanitize"; mkdir(dir.c_str(), 0755);
  std::mt19937_64 g(12345);
  // ---- index: transcripts that share exons (branching cDBG), one with a poly-A tail, a duplicate, a short one, two decoys
  std::vector<std::string> exons; for (int i = 0; i < 60; ++i) exons.push_back(rnd_seq(g, 40 + g() % 400));
  std::vector<std::string> names, seqs;
  for (int t = 0; t < 80; ++t) { std::string s; int ne = 2 + (int)(g() % 6); int e = (int)(g() % 50); for (int j = 0; j < ne; ++j) s += exons[(e + j * (1 + (int)(g() % 2))) % exons.size()]; if (t % 7 == 0) s += std::string(30, 'A'); names.push_back("tx" + std::to_string(t)); seqs.push_back(s); }
  names.push_back("dup"); seqs.push_back(seqs[3]); names.push_back("tiny"); seqs.push_back("ACGTACGTAC");
  const uint32_t first_decoy = (uint32_t)seqs.size();
  for (int d = 0; d < 2; ++d) { names.push_back("decoy" + std::to_string(d)); seqs.push_back(rnd_seq(g, 3000) + seqs[5 + d] + rnd_seq(g, 3000)); }
  std::vector<const char*> np, sp; std::vector<uint32_t> lens; for (size_t i = 0; i < seqs.size(); ++i) { np.push_back(names[i].c_str()); sp.push_back(seqs[i].data()); lens.push_back((uint32_t)seqs[i].size()); }
  sq_index_opts io; memset(&io, 0, sizeof(io)); io.threads = 4;
  sq_index* idx = nullptr; CHECK(sq_index_build_mem(&io, (uint32_t)seqs.size(), np.data(), sp.data(), lens.data(), first_decoy, (dir + "/idx").c_str(), &idx) == SQ_OK);
  sq_index_view v; CHECK(sq_index_get_view(idx, &v) == SQ_OK);
  // every k-mer of every kept reference must be found, on either strand
  uint64_t found = 0, tried = 0; const uint32_t k = v.k;
  for (uint32_t t = 0; t < v.num_refs; ++t) for (uint32_t p = 0; p + k <= v.ref_len[t]; p += 1 + (uint32_t)(g() % 3)) {
    uint64_t km = 0; for (uint32_t i = 0; i < k; ++i) { const uint64_t q = v.ref_accum[t] + p + i; km |= ((v.refseq[q >> 5] >> ((q & 31) * 2)) & 3ull) << (2 * i); }
    uint64_t u; uint32_t off; int fw; ++tried; found += sq_index_lookup_host(idx, km, &u, &off, &fw);
  }
  CHECK(found == tried && tried > 1000);
  sq_index* idx2 = nullptr; CHECK(sq_index_load((dir + "/idx").c_str(), -1, &idx2) == SQ_OK); CHECK(sq_index_num_kmers(idx2) == sq_index_num_kmers(idx)); sq_index_free(idx2);
  const uint32_t M = sq_index_first_decoy(idx);   // targets only: decoys are dropped before inference and output (SalmonQuantify.cpp:2479)
  // ---- reader: gzip FASTQ pairs + a wrapped FASTA, three batches in flight
  { gzFile f1 = gzopen((dir + "/r_1.fq.gz").c_str(), "wb"), f2 = gzopen((dir + "/r_2.fq.gz").c_str(), "wb");
    for (int i = 0; i < 5000; ++i) { std::string a = rnd_seq(g, 50 + g() % 120), b = rnd_seq(g, 50 + g() % 120);
      gzprintf(f1, "@r%d/1\n%s\n+\n%s\n", i, a.c_str(), std::string(a.size(), 'I').c_str()); gzprintf(f2, "@r%d/2\n%s\n+\n%s\n", i, b.c_str(), std::string(b.size(), 'I').c_str()); }
    gzclose(f1); gzclose(f2);
    const char* a1[] = {(dir + "/r_1.fq.gz").c_str()}; std::string p1 = dir + "/r_1.fq.gz", p2 = dir + "/r_2.fq.gz"; const char* q1[] = {p1.c_str()}; const char* q2[] = {p2.c_str()}; (void)a1;
    sq_reader* rd = nullptr; CHECK(sq_reader_open_ex(q1, 1, q2, 1, 700, 3, SQ_READER_KEEP_NAMES, &rd) == SQ_OK);   // names kept: the SAM writer's path
    uint64_t n = 0, bytes = 0; int held[2] = {-1, -1};
    for (;;) { sq_read_batch b; int slot; CHECK(sq_reader_next(rd, &b, &slot) == SQ_OK); if (b.n == 0) break;
      { const char* nm; const uint64_t* no; CHECK(sq_reader_names(rd, slot, &nm, &no) == SQ_OK);
        for (uint32_t i = 0; i < b.n; i += 131) { char want[32]; snprintf(want, sizeof want, "r%llu/1", (unsigned long long)(n + i)); CHECK(std::string(nm + no[i], nm + no[i + 1]) == want); } }
      n += b.n; bytes += b.seq_off[2 * b.n];
      for (uint64_t i = 0; i < b.seq_off[2 * b.n]; i += 97) CHECK(strchr("ACGT", (char)b.seq[i]) != nullptr);
      if (held[0] >= 0) sq_reader_release(rd, held[0]); held[0] = held[1]; held[1] = slot; }
    CHECK(n == 5000 && sq_reader_total(rd) == 5000 && bytes > 5000 * 100); sq_reader_close(rd); }
  // ---- reader: the same two gzip files cut into 64 KB pieces that the pool inflates in parallel (host/pgzip.cpp)
  { setenv("SQ_READER_PGZ_MIN", "1000", 1); setenv("SQ_READER_PGZ_PIECE", "65536", 1);
    std::string p1 = dir + "/r_1.fq.gz", p2 = dir + "/r_2.fq.gz"; const char* q1[] = {p1.c_str()}; const char* q2[] = {p2.c_str()};
    sq_reader* rd = nullptr; CHECK(sq_reader_open(q1, 1, q2, 1, 800, 3, &rd) == SQ_OK);
    uint64_t n = 0, bytes = 0;
    for (;;) { sq_read_batch b; int slot; CHECK(sq_reader_next(rd, &b, &slot) == SQ_OK); if (b.n == 0) break; n += b.n; bytes += b.seq_off[2 * b.n]; sq_reader_release(rd, slot); }
    CHECK(n == 5000 && bytes > 5000 * 100); sq_reader_close(rd);
    unsetenv("SQ_READER_PGZ_MIN"); unsetenv("SQ_READER_PGZ_PIECE"); }
  // ---- reader: the same reads as BGZF (64 KB gzip members inflated by the pool, taken in order by the stream thread)
  { auto bgzf = [&](const std::string& src, const std::string& dst) {
      std::string text; { gzFile f = gzopen(src.c_str(), "rb"); char buf[1 << 16]; int n; while ((n = gzread(f, buf, sizeof buf)) > 0) text.append(buf, (size_t)n); gzclose(f); }
      FILE* o = fopen(dst.c_str(), "wb");
      for (size_t i = 0; i <= text.size(); i += 50000) {   // the last round writes the empty end-of-file member when the text ends on a boundary or not
        const size_t len = i < text.size() ? std::min<size_t>(50000, text.size() - i) : 0;
        std::vector<unsigned char> cd(len + len / 100 + 1024); z_stream zs; memset(&zs, 0, sizeof zs); deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef*)(text.data() + i); zs.avail_in = (uInt)len; zs.next_out = cd.data(); zs.avail_out = (uInt)cd.size(); deflate(&zs, Z_FINISH); const size_t cl = zs.total_out; deflateEnd(&zs);
        const unsigned bs = (unsigned)(18 + cl + 8 - 1); const unsigned long crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef*)(text.data() + i), (uInt)len);
        const unsigned char hd[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bs & 0xff), (unsigned char)(bs >> 8)};
        fwrite(hd, 1, 18, o); fwrite(cd.data(), 1, cl, o);
        const unsigned char tl[8] = {(unsigned char)crc, (unsigned char)(crc >> 8), (unsigned char)(crc >> 16), (unsigned char)(crc >> 24), (unsigned char)len, (unsigned char)(len >> 8), (unsigned char)(len >> 16), (unsigned char)(len >> 24)};
        fwrite(tl, 1, 8, o);
        if (len == 0) break;
      }
      fclose(o); };
    bgzf(dir + "/r_1.fq.gz", dir + "/b_1.fq.gz"); bgzf(dir + "/r_2.fq.gz", dir + "/b_2.fq.gz");
    std::string p1 = dir + "/b_1.fq.gz", p2 = dir + "/b_2.fq.gz"; const char* q1[] = {p1.c_str()}; const char* q2[] = {p2.c_str()};
    sq_reader* rd = nullptr; CHECK(sq_reader_open(q1, 1, q2, 1, 900, 3, &rd) == SQ_OK);
    uint64_t n = 0;
    for (;;) { sq_read_batch b; int slot; CHECK(sq_reader_next(rd, &b, &slot) == SQ_OK); if (b.n == 0) break; n += b.n; sq_reader_release(rd, slot); }
    CHECK(n == 5000); sq_reader_close(rd);
    // [r4] damaged BGZF files (the own byte-mode inflate reads the members): bytes flipped anywhere, the file cut anywhere — an error or reads, never a crash
    { std::string whole; { FILE* f = fopen(p1.c_str(), "rb"); char buf[1 << 16]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) whole.append(buf, k); fclose(f); }
      int readable = 0;
      for (int it = 0; it < 120; ++it) { std::string b = whole;
        if (it % 4 == 0) b.resize(g() % b.size()); else for (int j = 0; j < 1 + (int)(g() % 6); ++j) b[g() % b.size()] = (char)(g() & 0xFF);
        const std::string mp = dir + "/mut_b.fq.gz"; FILE* f = fopen(mp.c_str(), "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f);
        const char* m1[] = {mp.c_str()}; sq_reader* r2 = nullptr; if (sq_reader_open(m1, 1, nullptr, 0, 700, 3, &r2) != SQ_OK) continue;
        bool ok = true; for (;;) { sq_read_batch rb; int slot; if (sq_reader_next(r2, &rb, &slot) != SQ_OK) { ok = false; break; } if (rb.n == 0) break; sq_reader_release(r2, slot); }
        readable += ok; sq_reader_close(r2); }
      printf("damaged BGZF: %d of 120 files read to the end\n", readable); } }
  // ---- eq classes inside gene-like groups -> normalizeAlphas (parallel union-find), twice, same answer
  std::vector<uint64_t> off{0}, cnt; std::vector<uint32_t> tid; std::vector<double> w;
  for (int c = 0; c < 4000; ++c) { uint32_t base = (uint32_t)(g() % (M - 8)); int n = 1 + (int)(g() % 5); std::vector<uint32_t> lab; for (int j = 0; j < n; ++j) lab.push_back(base + (uint32_t)(g() % 8)); std::sort(lab.begin(), lab.end()); lab.erase(std::unique(lab.begin(), lab.end()), lab.end());
    for (uint32_t t : lab) { tid.push_back(t); w.push_back(1.0 / (double)lab.size()); } off.push_back(tid.size()); cnt.push_back(1 + g() % 300); }
  sq_eq_table eq; memset(&eq, 0, sizeof(eq)); eq.num_classes = cnt.size(); eq.num_labels = tid.size(); eq.off = off.data(); eq.tid = tid.data(); eq.w = w.data(); eq.count = cnt.data();
  std::vector<double> lm(M), p1(M), p2(M); std::vector<uint64_t> uq(M), tc(M); std::vector<char> seen(M, 0); for (uint32_t t : tid) seen[t] = 1;
  for (uint32_t t = 0; t < M; ++t) { lm[t] = seen[t] ? std::log(1e-3 + (double)(g() % 1000)) : HUGE_VAL; tc[t] = g() % 500; uq[t] = tc[t] / 3; }
  CHECK(sq_normalize_alphas(M, &eq, lm.data(), uq.data(), tc.data(), p1.data()) == SQ_OK); CHECK(sq_normalize_alphas(M, &eq, lm.data(), uq.data(), tc.data(), p2.data()) == SQ_OK);
  CHECK(memcmp(p1.data(), p2.data(), M * 8) == 0);
  // ---- writers and the eq-class file round trip
  std::vector<double> eff(M); for (uint32_t t = 0; t < M; ++t) eff[t] = std::max(1.0, (double)sq_index_ref_len(idx, t) - 200.0);
  mkdir((dir + "/out").c_str(), 0755); mkdir((dir + "/out/aux_info").c_str(), 0755);
  CHECK(sq_write_quant_sf((dir + "/out/quant.sf").c_str(), idx, eff.data(), p1.data(), 0.0) == SQ_OK);
  CHECK(sq_write_eq_classes((dir + "/out/aux_info/eq_classes.txt.gz").c_str(), idx, &eq, 1) == SQ_OK);
  CHECK(sq_write_ambig_info((dir + "/out/aux_info/ambig_info.tsv").c_str(), M, &eq) == SQ_OK);
  uint64_t lc[64]; for (auto& x : lc) x = g() % 1000; CHECK(sq_write_lib_format_counts((dir + "/out/lib_format_counts.json").c_str(), "r_1.fq.gz,r_2.fq.gz", 1, 2, 4, lc, 12345, 12000) == SQ_OK);
  sq_eq_file* ef = nullptr; CHECK(sq_eq_file_read((dir + "/out/aux_info/eq_classes.txt.gz").c_str(), &ef) == SQ_OK); CHECK(sq_eq_file_num_txp(ef) == M);
  sq_eq_table back; memset(&back, 0, sizeof(back)); CHECK(sq_eq_file_table(ef, &back) == SQ_OK); CHECK(back.num_classes == eq.num_classes && back.num_labels == eq.num_labels && memcmp(back.tid, eq.tid, eq.num_labels * 4) == 0);
  sq_eq_file_free(ef);
  std::vector<const char*> nm; for (uint32_t t = 0; t < M; ++t) nm.push_back(sq_index_ref_name(idx, t));
  sq_boot_writer* bw = nullptr; CHECK(sq_boot_writer_open((dir + "/out/aux_info").c_str(), M, nm.data(), &bw) == SQ_OK); for (int r = 0; r < 5; ++r) CHECK(sq_boot_writer_append(bw, p1.data(), M) == SQ_OK); CHECK(sq_boot_writer_close(bw) == 5);
  // ---- malformed input: mutated / truncated copies of a small eq-class file and of a small FASTQ must be rejected or read, never crash
  { std::vector<uint64_t> o2(off.begin(), off.begin() + 101), c2(cnt.begin(), cnt.begin() + 100); sq_eq_table small = eq; small.num_classes = 100; small.num_labels = o2[100]; small.off = o2.data(); small.count = c2.data();
    const std::string src = dir + "/small_eq.txt", mut = dir + "/mut_eq.txt";
    CHECK(sq_write_eq_classes(src.c_str(), idx, &small, 1) == SQ_OK);
    std::string bytes; { gzFile f = gzopen(src.c_str(), "rb"); char buf[4096]; int n; while ((n = gzread(f, buf, sizeof buf)) > 0) bytes.append(buf, (size_t)n); gzclose(f); }
    int accepted = 0;
    for (int it = 0; it < 300; ++it) { std::string b = bytes; const int nm = 1 + (int)(g() % 4);
      for (int j = 0; j < nm; ++j) { const size_t p = g() % b.size(); const int op = (int)(g() % 4);
        if (op == 0) b[p] = (char)(g() & 0xFF); else if (op == 1) b.resize(p); else if (op == 2) b.insert(p, std::to_string(g())); else b[p] = "0123456789 \n\t-e."[g() % 16];
        if (b.empty()) b = "1"; }
      FILE* f = fopen(mut.c_str(), "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f);
      sq_eq_file* e2 = nullptr; if (sq_eq_file_read(mut.c_str(), &e2) == SQ_OK) { sq_eq_table t2; memset(&t2, 0, sizeof(t2)); CHECK(sq_eq_file_table(e2, &t2) == SQ_OK); for (uint64_t i = 0; i < t2.num_labels; ++i) CHECK(t2.tid[i] < sq_eq_file_num_txp(e2)); sq_eq_file_free(e2); ++accepted; } }
    std::string fq; for (int i = 0; i < 40; ++i) { std::string a = rnd_seq(g, 30 + g() % 60); fq += "@r" + std::to_string(i) + "\n" + a + "\n+\n" + std::string(a.size(), 'I') + "\n"; }
    int ok_reads = 0;
    for (int it = 0; it < 300; ++it) { std::string b = fq; const int nm = 1 + (int)(g() % 4);
      for (int j = 0; j < nm; ++j) { const size_t p = g() % b.size(); const int op = (int)(g() % 4);
        if (op == 0) b[p] = (char)(g() & 0xFF); else if (op == 1) b.resize(p); else if (op == 2) b.insert(p, "\n@x\n"); else b[p] = "@+>\n\rACGTN"[g() % 10];
        if (b.empty()) b = "@"; }
      const std::string mp = dir + "/mut.fq"; FILE* f = fopen(mp.c_str(), "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f);
      const char* q1[] = {mp.c_str()}; sq_reader* rd = nullptr; CHECK(sq_reader_open_ex(q1, 1, nullptr, 0, 16, 2, (it & 1) ? SQ_READER_KEEP_NAMES : 0u, &rd) == SQ_OK);
      for (;;) { sq_read_batch rb; int slot; if (sq_reader_next(rd, &rb, &slot) != SQ_OK || rb.n == 0) break; CHECK(rb.seq_off[rb.n] < (1u << 20)); ok_reads += (int)rb.n;
        if (it & 1) { const char* nm; const uint64_t* no; CHECK(sq_reader_names(rd, slot, &nm, &no) == SQ_OK); CHECK(no[rb.n] < (1u << 20)); }
        sq_reader_release(rd, slot); }
      sq_reader_close(rd); }
    // corrupt index files: truncated anywhere, or with bytes of the header / section tables overwritten
    { std::string ib; { FILE* f = fopen((dir + "/idx/index.bin").c_str(), "rb"); CHECK(f != nullptr); char buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) ib.append(buf, n); fclose(f); }
      mkdir((dir + "/idx_mut").c_str(), 0755); int loaded = 0;
      for (int it = 0; it < 200; ++it) { std::string b = ib;
        if (it % 3 == 0) b.resize(g() % b.size()); else for (int j = 0; j < 1 + (int)(g() % 3); ++j) b[g() % std::min<size_t>(b.size(), it % 3 == 1 ? 512 : b.size())] = (char)(g() & 0xFF);
        FILE* f = fopen((dir + "/idx_mut/index.bin").c_str(), "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f);
        sq_index* bad = nullptr; if (sq_index_load((dir + "/idx_mut").c_str(), -1, &bad) == SQ_OK) { ++loaded; sq_index_free(bad); } }
      printf("corrupt index: %d of 200 damaged files still loaded\n", loaded); }
    printf("malformed input: %d of 300 mutated eq files still parsed, %d reads came out of 300 mutated FASTQ files\n", accepted, ok_reads); }
  // ---- [r4] alignment-based input: a name-grouped SAM file of pairs, orphans and unmapped reads through sq_sam_*; then mutated copies (never a crash)
  { std::string sam = "@HD\tVN:1.6\tSO:unsorted\n"; for (uint32_t t = 0; t < M; ++t) sam += "@SQ\tSN:" + std::string(sq_index_ref_name(idx, t)) + "\tLN:" + std::to_string(sq_index_ref_len(idx, t)) + "\n";
    sam += "@PG\tID:x\n"; uint64_t want_frags = 0;
    for (int i = 0; i < 600; ++i) { const std::string nm = "q" + std::to_string(i); const int kind = (int)(g() % 10); const int nal = 1 + (int)(g() % 3);
      if (kind == 0) { sam += nm + "\t77\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\n" + nm + "\t141\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\n"; continue; }
      ++want_frags;
      for (int a = 0; a < nal; ++a) { const uint32_t t = (uint32_t)(g() % M); const uint32_t L = sq_index_ref_len(idx, t); const int p1 = 1 + (int)(g() % (L > 400 ? L - 400 : 1)), fl = 150 + (int)(g() % 200);
        const std::string rn = sq_index_ref_name(idx, t); const int as = -(int)(g() % 30); const int sec = a ? 256 : 0;
        if (kind == 1) sam += nm + "\t" + std::to_string(73 + sec) + "\t" + rn + "\t" + std::to_string(p1) + "\t1\t100M\t*\t0\t0\t*\t*\tAS:i:" + std::to_string(as) + "\n";      // mate unmapped: an orphan
        else { sam += nm + "\t" + std::to_string(99 + sec) + "\t" + rn + "\t" + std::to_string(p1) + "\t1\t100M\t=\t" + std::to_string(p1 + fl - 100) + "\t" + std::to_string(fl) + "\t*\t*\tAS:i:" + std::to_string(as) + "\tNH:i:" + std::to_string(nal) + "\n";
               sam += nm + "\t" + std::to_string(147 + sec) + "\t" + rn + "\t" + std::to_string(p1 + fl - 100) + "\t1\t100M\t=\t" + std::to_string(p1) + "\t-" + std::to_string(fl) + "\t*\t*\tAS:i:" + std::to_string(as) + "\n"; } } }
    const std::string sp = dir + "/a.sam"; { FILE* f = fopen(sp.c_str(), "wb"); fwrite(sam.data(), 1, sam.size(), f); fclose(f); }
    auto drain = [&](const std::string& path, bool must_open, uint64_t* frags) -> bool {
      sq_sam* sm = nullptr; const int rc = sq_sam_open(path.c_str(), 1, &sm); if (rc != SQ_OK) { CHECK(!must_open); return false; }
      const uint32_t nr = sq_sam_num_refs(sm); for (uint32_t i = 0; i < nr && i < 3; ++i) { CHECK(sq_sam_ref_name(sm, i) != nullptr); (void)sq_sam_ref_len(sm, i); }
      bool ok = true; *frags = 0;
      for (;;) { sq_aln_batch ab; memset(&ab, 0, sizeof ab); sq_sam_counts sc; const int r2 = sq_sam_next(sm, 97, (int)(g() & 1), 1.0, &ab, &sc); if (r2 != SQ_OK) { ok = false; break; } if (!ab.n) break;
        *frags += ab.n; CHECK(ab.read_off[0] == 0);
        for (uint32_t i = 0; i < ab.n; ++i) { CHECK(ab.read_off[i] <= ab.read_off[i + 1]); for (uint64_t a = ab.read_off[i]; a < ab.read_off[i + 1]; ++a) CHECK(ab.aln[a].tid < nr); } }
      sq_sam_close(sm); return ok; };
    uint64_t frags = 0; CHECK(drain(sp, true, &frags)); CHECK(frags == want_frags);
    int sam_ok = 0;
    for (int it = 0; it < 300; ++it) { std::string b = sam; const int nm = 1 + (int)(g() % 4);
      for (int j = 0; j < nm; ++j) { const size_t p = g() % b.size(); const int op = (int)(g() % 5);
        if (op == 0) b[p] = (char)(g() & 0xFF); else if (op == 1) b.resize(p); else if (op == 2) b.insert(p, std::to_string(g())); else if (op == 3) b[p] = "\t\n@*=-0123456789"[g() % 16]; else b.insert(p, "\t");
        if (b.empty()) b = "@"; }
      const std::string mp = dir + "/mut.sam"; FILE* f = fopen(mp.c_str(), "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f);
      uint64_t fr = 0; sam_ok += drain(mp, false, &fr) ? 1 : 0; }
    printf("malformed SAM: %d of 300 mutated files read to the end\n", sam_ok);
    // the same records as BAM (binary: SAM spec 4.2; written here field by field), then damaged copies of the byte stream
    { std::string bamb("BAM\1", 4); auto put32 = [&](std::string& o, int32_t v) { for (int k = 0; k < 4; ++k) o.push_back((char)((uint32_t)v >> (8 * k))); };
      std::string text; std::vector<std::pair<std::string, uint32_t>> refs; std::vector<std::vector<std::string>> recs;
      { size_t p0 = 0; while (p0 < sam.size()) { const size_t nl = sam.find('\n', p0); std::string ln = sam.substr(p0, nl - p0); p0 = nl + 1;
          if (ln[0] == '@') { text += ln + "\n"; if (!ln.compare(0, 3, "@SQ")) { const size_t a = ln.find("SN:"), b = ln.find("\tLN:"); refs.push_back({ln.substr(a + 3, b - a - 3), (uint32_t)atoi(ln.c_str() + b + 4)}); } continue; }
          std::vector<std::string> f; size_t q = 0; for (;;) { const size_t t = ln.find('\t', q); f.push_back(ln.substr(q, t == std::string::npos ? std::string::npos : t - q)); if (t == std::string::npos) break; q = t + 1; } recs.push_back(f); } }
      put32(bamb, (int32_t)text.size()); bamb += text; put32(bamb, (int32_t)refs.size());
      std::unordered_map<std::string, int32_t> rid; for (size_t i = 0; i < refs.size(); ++i) { rid[refs[i].first] = (int32_t)i; put32(bamb, (int32_t)refs[i].first.size() + 1); bamb += refs[i].first; bamb.push_back('\0'); put32(bamb, (int32_t)refs[i].second); }
      for (auto& f : recs) { std::string body; const int32_t ref = rid.count(f[2]) ? rid[f[2]] : -1, mref = f[6] == "=" ? ref : (rid.count(f[6]) ? rid[f[6]] : -1);
        std::vector<uint32_t> cg; { uint32_t num = 0; for (char ch : f[5]) { if (ch >= '0' && ch <= '9') num = num * 10 + (uint32_t)(ch - '0'); else if (ch != '*') { cg.push_back((num << 4) | (uint32_t)(std::string("MIDNSHP=X").find(ch))); num = 0; } } }
        const int32_t lseq = f[9] == "*" ? 0 : (int32_t)f[9].size();
        put32(body, ref); put32(body, atoi(f[3].c_str()) - 1); body.push_back((char)(f[0].size() + 1)); body.push_back((char)atoi(f[4].c_str())); body.push_back(0x48); body.push_back(0x12);
        body.push_back((char)(cg.size() & 0xFF)); body.push_back((char)(cg.size() >> 8)); const int fl = atoi(f[1].c_str()); body.push_back((char)(fl & 0xFF)); body.push_back((char)(fl >> 8));
        put32(body, lseq); put32(body, mref); put32(body, atoi(f[7].c_str()) - 1); put32(body, atoi(f[8].c_str()));
        body += f[0]; body.push_back('\0'); for (uint32_t c : cg) put32(body, (int32_t)c); body.append((size_t)(lseq + 1) / 2, (char)0x11); body.append((size_t)lseq, (char)0xFF);
        for (size_t t = 11; t < f.size(); ++t) if (!f[t].compare(0, 5, "AS:i:")) { body += "ASi"; put32(body, atoi(f[t].c_str() + 5)); } else if (!f[t].compare(0, 5, "NH:i:")) { body += "NHC"; body.push_back((char)atoi(f[t].c_str() + 5)); }
        put32(bamb, (int32_t)body.size()); bamb += body; }
      auto write_gz = [&](const std::string& path, const std::string& bytes) { gzFile o = gzopen(path.c_str(), "wb1"); gzwrite(o, bytes.data(), (unsigned)bytes.size()); gzclose(o); };
      const std::string bp = dir + "/a.bam"; write_gz(bp, bamb);
      uint64_t fb = 0; CHECK(drain(bp, true, &fb)); CHECK(fb == want_frags);
      int bam_ok = 0;
      for (int it = 0; it < 300; ++it) { std::string b = bamb; const int nm = 1 + (int)(g() % 4);
        for (int j = 0; j < nm; ++j) { const size_t p = g() % b.size(); const int op = (int)(g() % 3); if (op == 0) b[p] = (char)(g() & 0xFF); else if (op == 1) b.resize(p); else b.insert(p, std::string(1 + g() % 8, (char)(g() & 0xFF))); if (b.empty()) b = "B"; }
        const std::string mp = dir + "/mut.bam"; write_gz(mp, b); uint64_t fr = 0; bam_ok += drain(mp, false, &fr) ? 1 : 0; }
      printf("malformed BAM: %d of 300 damaged byte streams read to the end\n", bam_ok); } }
  // ---- [r4] the rest of the output directory: meta_info.json, the index digests, fld.gz, the bias dumps
  { const std::string aux = dir + "/out/aux_info";
    for (int wch = 0; wch < 6; ++wch) { const char* h = sq_index_hash(idx, wch); CHECK(h != nullptr); }
    std::vector<double> lp(1001); for (int i = 0; i <= 1000; ++i) lp[i] = -0.5 * ((i - 250.0) / 25.0) * ((i - 250.0) / 25.0) - 4.0;
    double mean = 0, sd = 0; uint32_t sup = 0; CHECK(sq_write_fld_samples((aux + "/fld.gz").c_str(), lp.data(), 1, 1000, 10000, 7, &mean, &sd, &sup) == SQ_OK); CHECK(sup == 1001 && mean > 200 && mean < 300 && sd > 10 && sd < 40);
    uint32_t nbins = 0; CHECK(sq_write_legacy_bias(aux.c_str(), &nbins) == SQ_OK); CHECK(nbins > 0);
    std::vector<double> tot(3, 1.0), cn(75, 0.5); CHECK(sq_write_gc_model((aux + "/obs_gc.gz").c_str(), 0, 3, 25, tot.data(), cn.data()) == SQ_OK);
    std::vector<double> sm(9 * 64, -1.386); CHECK(sq_write_seq_model((aux + "/obs5_seq.gz").c_str(), sm.data()) == SQ_OK);
    const uint32_t lb[5] = {791, 1265, 1707, 2433, 0xFFFFFFFFu}; std::vector<double> pm(5 * 20, 0.05); CHECK(sq_write_pos_models((aux + "/obs5_pos.gz").c_str(), 5, lb, 20, pm.data()) == SQ_OK);
    sq_meta_info mi; memset(&mi, 0, sizeof mi); const char* lt[] = {"IU"}; const uint32_t lcl[5] = {791, 1265, 1707, 2433, 100000};
    mi.samp_type = "none"; mi.opt_type = "vb"; mi.num_libraries = 1; mi.library_types = lt; mi.frag_dist_length = 1001; mi.frag_length_mean = mean; mi.frag_length_sd = sd; mi.num_bias_bins = nbins;
    mi.mapping_type = "mapping"; mi.keep_duplicates = 0; mi.num_valid_targets = M; mi.num_eq_classes = cnt.size(); mi.num_length_classes = 5; mi.length_classes = lcl;
    mi.index_seq_hash = sq_index_hash(idx, 0); mi.index_name_hash = sq_index_hash(idx, 1); mi.index_seq_hash512 = sq_index_hash(idx, 2); mi.index_name_hash512 = sq_index_hash(idx, 3);
    mi.index_decoy_seq_hash = sq_index_hash(idx, 4); mi.index_decoy_name_hash = sq_index_hash(idx, 5);
    mi.num_processed = 5000; mi.num_mapped = 4900; mi.percent_mapped = 98.0; mi.start_time = "Thu Sep 24 12:00:00 2026"; mi.end_time = "Thu Sep 24 12:00:01 2026"; mi.backend = "gfx950"; mi.runtime_s = 1.0;
    CHECK(sq_write_meta_info((aux + "/meta_info.json").c_str(), &mi) == SQ_OK);
    mi.quant_errors = "a \"quoted\" reason\n"; mi.keep_duplicates = -1; mi.library_types = nullptr; mi.num_libraries = 0; CHECK(sq_write_meta_info((aux + "/meta_info_err.json").c_str(), &mi) == SQ_OK); }
  sq_index_free(idx);
  printf("host sanitize run ok: %u refs, %llu k-mer lookups, 5000 read pairs, %zu classes\n", M, (unsigned long long)tried, cnt.size());
  return 0;
}
