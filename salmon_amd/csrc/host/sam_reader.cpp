// host/sam_reader.cpp — [r4] the record source of alignment-based mode (`salmon quant -a`): a name-collated SAM text file (plain or gzip) read
// into the alignment records the online stage consumes (sq_aln), fragment by fragment.  Follows the reference's BAMQueue
// (include/salmon/internal/alignment/BAMQueue.tpp:288-343 getPairedAlignmentType_, :355-545 getFrag_(ReadPair), :548-600 getFrag_(UnpairedRead)),
// ReadPair / UnpairedRead (ReadPair.hpp:61-200: pos, fwd, fragLen, getAS, mateStatus) and salmon::utils::hitType (SalmonUtils.cpp:531-652); the
// reference reads BAM/SAM through htslib (staden io_lib), which is not available here — an own parser takes its place: SAM text (plain or gzip), and
// [r4] BAM (the BGZF members are inflated in parallel by bgzf_source.h — or, for a plain gzip stream, through zlib's gzip reader; header and records are
// decoded here: SAM spec section 4.2).
//   paired library: a record whose read and mate are mapped, flagged proper pair, on the same target -> a pair together with the NEXT record;
//     read mapped, mate not (or not a proper pair, or another target) -> an orphan alignment (left if FREAD1 else right);
//     read unmapped -> skipped; both unmapped -> an unaligned fragment (counted).
//   consecutive alignments with the same read name are one fragment; its alignments are ordered by transcript (AlignmentGroup::sortHits).
// The error model of alignment mode (AlignmentModel.hpp, learned from CIGAR strings) is NOT built: the conditional probability of an alignment is
// either 1 (`--noErrorModel`) or exp(-scoreExp (bestAS - AS)) from the AS tags (`--useASWithoutCIGAR`, SalmonQuantifyAlignments.cpp:516-521).
#include "index.h"
#include "bgzf_source.h"
#include <zlib.h>
#include <thread>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
enum { F_PAIRED = 1, F_PROPER = 2, F_UNMAP = 4, F_MUNMAP = 8, F_REVERSE = 16, F_READ1 = 64, F_READ2 = 128 };
struct Rec { std::string name; int flag = 0; int32_t ref = -1, mref = -1, pos = 0; uint32_t len = 0; int32_t as = 0; bool has_as = false;
  std::vector<uint32_t> cig; std::vector<uint8_t> seq; };   // [r5] kept on request (sq_sam_keep_reads): BAM-encoded CIGAR operations, bases as 0..3 (anything but ACGT: 0, as the reference's samToTwoBit maps them)
// what the error model needs of an alignment: the record scored with the left matrices [0] and the one scored with the right ones [1]
struct Side { int32_t pos[2] = {0, 0}; std::vector<uint32_t> cig[2]; std::vector<uint8_t> seq[2]; uint8_t mate[2] = {0, 0} /* 1 = first read, 2 = second */, rev[2] = {0, 0}; int32_t as_sum = 0;
  void put(int k, Rec& r) { pos[k] = r.pos; cig[k] = std::move(r.cig); seq[k] = std::move(r.seq); mate[k] = (r.flag & F_READ2) ? 2 : 1; rev[k] = (r.flag & F_REVERSE) ? 1 : 0; }
  void clear() { for (int k = 0; k < 2; ++k) { pos[k] = 0; cig[k].clear(); seq[k].clear(); mate[k] = rev[k] = 0; } as_sum = 0; } };
inline uint8_t fmt_id(uint8_t t, uint8_t o, uint8_t s) { return (uint8_t)(t | (o << 1) | (s << 3)); }
// hitType(end1Start, end1Fwd, end2Start, end2Fwd) — SalmonUtils.cpp:531-575 (the four-argument form: no dovetail stretch)
inline uint8_t hit_type_pair(int32_t s1, bool f1, int32_t s2, bool f2) {
  if (f1 != f2) { if (f1) return s1 <= s2 ? fmt_id(1, 2, 0) : fmt_id(1, 1, 0); return s2 <= s1 ? fmt_id(1, 2, 1) : fmt_id(1, 1, 1); }
  return f1 ? fmt_id(1, 0, 2) : fmt_id(1, 0, 3);
}
inline uint8_t hit_type_single(bool fwd) { return fwd ? fmt_id(0, 3, 2) : fmt_id(0, 3, 3); }   // :638-646
}  // namespace

struct sq_sam {
  gzFile f = nullptr; std::string path; bool paired = true;
  // [r4] a BGZF file (every BAM, a bgzipped SAM) is inflated by a small pool, member groups in parallel (bgzf_source.h), and handed over as one byte
  // stream; anything else goes through zlib's gzip reader (plain text too)
  std::unique_ptr<sqio::Pool> pool; std::unique_ptr<sqio::BgzfSource> bg; PgzBuf cur; size_t cur_off = 0; std::string io_err;
  ~sq_sam() { cur = PgzBuf(); bg.reset(); pool.reset(); }   // the source's tasks run on the pool: the source goes first
  int fill(char* dst, size_t cap) {   // > 0 bytes, 0 at the end, < 0 on error (io_err)
    if (!bg) {   // zlib's reader: a damaged or truncated stream is an error, not a shorter file
      const int n = gzread(f, dst, (unsigned)cap);
      if (n <= 0) { int e = Z_OK; const char* msg = gzerror(f, &e); if (n < 0 || (e != Z_OK && e != Z_STREAM_END)) { io_err = std::string("damaged or truncated compressed stream (") + (msg ? msg : "zlib") + ")"; return -1; } }
      return n;
    }
    for (;;) {
      if (cur_off < cur.n) { const size_t take = std::min(cap, cur.n - cur_off); memcpy(dst, cur.p + cur_off, take); cur_off += take; return (int)take; }
      cur = PgzBuf(); cur_off = 0; const int rc = bg->next_buf(&cur); if (rc < 0) { io_err = bg->err; return -1; } if (rc == 0) return 0;
    }
  }
  std::vector<std::string> names; std::vector<uint32_t> lens; std::unordered_map<std::string, int32_t> by_name;
  std::vector<uint32_t> tid_map;
  std::vector<char> buf; size_t bpos = 0, bend = 0; bool eof = false;
  std::string line; bool have_pending_line = false;
  // the alignment that opened the next fragment (read ahead)
  bool have_next = false; sq_aln next_aln{}; std::string next_name; int32_t next_as = 0; bool next_has_as = false;
  sq_sam_counts cnt{};
  // output arrays of the current batch
  std::vector<uint64_t> off; std::vector<sq_aln> alns; std::vector<int32_t> as_of; std::vector<uint8_t> has_as_of;
  // [r5] the reads behind the alignments, for the CIGAR-based error model (AlignmentModel.cpp): per alignment two records (left / right matrices)
  bool keep_reads = false, bowtie2 = false; std::vector<Side> sides; Side next_side; std::vector<uint8_t> frag_seq[2];
  std::vector<uint64_t> r_cig_off, r_seq_off; std::vector<uint32_t> r_cig; std::vector<uint8_t> r_seq; std::vector<int32_t> r_pos, r_score;
  bool getline(std::string& out) {
    out.clear();
    for (;;) {
      if (bpos == bend) { if (eof) return !out.empty(); const int n = fill(buf.data(), buf.size()); if (n <= 0) { eof = true; return !out.empty(); } bpos = 0; bend = (size_t)n; }
      const char* s = buf.data() + bpos; const char* e = (const char*)memchr(s, '\n', bend - bpos);
      if (e) { out.append(s, e - s); bpos = (size_t)(e - buf.data()) + 1; if (!out.empty() && out.back() == '\r') out.pop_back(); return true; }
      out.append(s, bend - bpos); bpos = bend;
    }
  }
  // ---- BAM (binary; the decompressed stream): little-endian fields
  bool bam = false;
  bool read_exact(void* dst, size_t n) {
    char* d = (char*)dst;
    while (n) {
      if (bpos == bend) { if (eof) return false; const int k = fill(buf.data(), buf.size()); if (k <= 0) { eof = true; return false; } bpos = 0; bend = (size_t)k; }
      const size_t take = std::min(n, bend - bpos); memcpy(d, buf.data() + bpos, take); d += take; bpos += take; n -= take;
    }
    return true;
  }
  static int32_t i32(const uint8_t* p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
  std::vector<uint8_t> blk;
  bool next_record_bam(Rec& r, std::string& err) {
    uint8_t b4[4];
    if (!read_exact(b4, 4)) return false;                       // the end of the file
    const int32_t bs = i32(b4);
    if (bs < 32 || bs > (64 << 20)) { err = "malformed BAM record (block size " + std::to_string(bs) + ")"; return false; }
    blk.resize((size_t)bs);
    if (!read_exact(blk.data(), (size_t)bs)) { err = "truncated BAM file (inside a record)"; return false; }
    const uint8_t* p = blk.data();
    const int32_t ref = i32(p), pos = i32(p + 4); const uint32_t l_name = p[8]; const uint32_t n_cig = (uint32_t)p[12] | ((uint32_t)p[13] << 8); const uint32_t flag = (uint32_t)p[14] | ((uint32_t)p[15] << 8);
    const int32_t l_seq = i32(p + 16), mref = i32(p + 20);
    const size_t fixed = 32, need = fixed + l_name + (size_t)n_cig * 4 + ((size_t)std::max(l_seq, 0) + 1) / 2 + (size_t)std::max(l_seq, 0);
    if (l_seq < 0 || l_name == 0 || need > (size_t)bs) { err = "malformed BAM record (field lengths exceed the block)"; return false; }
    r.name.assign((const char*)p + fixed, strnlen((const char*)p + fixed, l_name));
    r.flag = (int)flag; r.ref = (ref >= 0 && (size_t)ref < names.size()) ? ref : -1; r.mref = (mref >= 0 && (size_t)mref < names.size()) ? mref : -1; r.pos = pos;
    if (keep_reads) {
      const uint8_t* cg = p + fixed + l_name; r.cig.resize(n_cig); for (uint32_t i = 0; i < n_cig; ++i) r.cig[i] = (uint32_t)i32(cg + 4 * i);
      const uint8_t* sq = cg + (size_t)n_cig * 4; r.seq.resize((size_t)l_seq);
      for (int32_t i = 0; i < l_seq; ++i) { const uint8_t c = (sq[i >> 1] >> ((~i & 1) << 2)) & 15u; r.seq[(size_t)i] = c == 2 ? 1 : c == 4 ? 2 : c == 8 ? 3 : 0; }
    }
    if (l_seq > 0) r.len = (uint32_t)l_seq;
    else { uint32_t L = 0; const uint8_t* c = p + fixed + l_name; for (uint32_t i = 0; i < n_cig; ++i) { const uint32_t v = (uint32_t)i32(c + 4 * i); const uint32_t op = v & 15u; if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) L += v >> 4; } r.len = L; }   // M I S = X consume the query
    // the optional fields: TAG (2) TYPE (1) VALUE; AS is an integer of whatever width the writer chose
    r.has_as = false; r.as = 0;
    const uint8_t* a = p + need; const uint8_t* e = p + bs;
    while (a + 3 <= e) {
      const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2]; a += 3; size_t w = 0; long long iv = 0; bool isint = true;
      switch (ty) {
        case 'A': w = 1; isint = false; break;
        case 'c': w = 1; if (a + 1 <= e) iv = (int8_t)a[0]; break;
        case 'C': w = 1; if (a + 1 <= e) iv = a[0]; break;
        case 's': w = 2; if (a + 2 <= e) iv = (int16_t)((uint16_t)a[0] | ((uint16_t)a[1] << 8)); break;
        case 'S': w = 2; if (a + 2 <= e) iv = (uint16_t)((uint16_t)a[0] | ((uint16_t)a[1] << 8)); break;
        case 'i': w = 4; if (a + 4 <= e) iv = i32(a); break;
        case 'I': w = 4; if (a + 4 <= e) iv = (uint32_t)i32(a); break;
        case 'f': w = 4; isint = false; break;
        case 'Z': case 'H': { const uint8_t* z = (const uint8_t*)memchr(a, 0, (size_t)(e - a)); if (!z) { a = e; w = 0; } else w = (size_t)(z - a) + 1; isint = false; break; }
        case 'B': { if (a + 5 > e) { a = e; w = 0; break; } const char st = (char)a[0]; const uint32_t cnt2 = (uint32_t)i32(a + 1); const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; w = 5 + (size_t)cnt2 * es; isint = false; break; }
        default: a = e; w = 0; isint = false; break;   // an unknown type: nothing behind it can be found
      }
      if (a + w > e) break;
      if (t0 == 'A' && t1 == 'S' && isint) { r.as = (int32_t)iv; r.has_as = true; }
      a += w;
    }
    if (r.ref < 0 && !(r.flag & F_UNMAP)) r.flag |= F_UNMAP;
    cnt.num_records++;
    return true;
  }
  bool next_record(Rec& r, std::string& err) {
    if (bam) return next_record_bam(r, err);
    while (getline(line)) {
      if (line.empty() || line[0] == '@') continue;
      // QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL [TAG...]
      const char* p = line.c_str(); const char* fld[12]; int nf = 0; fld[nf++] = p;
      std::vector<const char*> tags;
      for (const char* q = p; *q; ++q) if (*q == '\t') { if (nf < 11) fld[nf++] = q + 1; else tags.push_back(q + 1); }
      if (nf < 11) { err = "malformed SAM record (fewer than 11 fields): " + line.substr(0, 80); return false; }
      auto field = [&](int i) { const char* a = fld[i]; const char* b = (i + 1 < nf) ? fld[i + 1] - 1 : (tags.empty() ? p + line.size() : tags[0] - 1); return std::string(a, b - a); };
      r.name = field(0); r.flag = atoi(fld[1]);
      const std::string rn = field(2), rnext = field(6);
      auto it = by_name.find(rn); r.ref = (rn == "*" || it == by_name.end()) ? -1 : it->second;
      if (rnext == "=") r.mref = r.ref; else { auto it2 = by_name.find(rnext); r.mref = (rnext == "*" || it2 == by_name.end()) ? -1 : it2->second; }
      r.pos = atoi(fld[3]) - 1;
      const std::string seq = field(9);
      if (keep_reads) {
        r.cig.clear(); r.seq.clear(); const std::string cg = field(5);
        if (cg != "*") { uint32_t num = 0; for (char ch : cg) { if (ch >= '0' && ch <= '9') num = num * 10 + (uint32_t)(ch - '0'); else { const char* ops = "MIDNSHP=X"; const char* w = strchr(ops, ch); r.cig.push_back((num << 4) | (uint32_t)(w ? w - ops : 15)); num = 0; } } }
        if (seq != "*") { r.seq.resize(seq.size()); for (size_t i = 0; i < seq.size(); ++i) { const char ch = (char)(seq[i] & ~0x20); r.seq[i] = ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 0; } }
      }
      if (seq != "*") r.len = (uint32_t)seq.size();
      else {   // no sequence stored (secondary records): the read length is what the CIGAR consumes of the query
        const std::string cg = field(5); uint32_t num = 0, L = 0;
        for (char ch : cg) { if (ch >= '0' && ch <= '9') num = num * 10 + (uint32_t)(ch - '0'); else { if (ch == 'M' || ch == 'I' || ch == 'S' || ch == '=' || ch == 'X') L += num; num = 0; } }
        r.len = L;
      }
      r.has_as = false; r.as = 0;
      for (size_t t = 0; t < tags.size(); ++t) if (!strncmp(tags[t], "AS:i:", 5)) { r.as = atoi(tags[t] + 5); r.has_as = true; break; }
      if (r.ref < 0 && !(r.flag & F_UNMAP)) { r.flag |= F_UNMAP; }   // a target that is not in the header cannot be used
      cnt.num_records++;
      return true;
    }
    return false;
  }
  static std::string base_name(const std::string& n) { if (n.size() > 2 && n[n.size() - 2] == '/') return n.substr(0, n.size() - 2); return n; }   // ReadPair::getNameLength
  // the next alignment (a pair or an orphan / single read) or false at the end of the file
  bool next_alignment(sq_aln& a, std::string& name, int32_t& as, bool& has_as, std::string& err, Side& sd) {
    Rec r1, r2; sd.clear();
    for (;;) {
      if (!next_record(r1, err)) return false;
      const bool mapped = !(r1.flag & F_UNMAP);
      if (!paired) {   // getFrag_(UnpairedRead): every mapped record is an alignment
        if (!mapped) { cnt.num_unaligned++; continue; }
        memset(&a, 0, sizeof(a)); a.tid = (uint32_t)r1.ref; a.pos = r1.pos; a.fwd = !(r1.flag & F_REVERSE); a.read_len = (uint16_t)std::min<uint32_t>(r1.len, 65535u);
        a.mate_status = SQ_MS_SINGLE_END; a.format_id = hit_type_single(a.fwd); a.score = r1.as; a.est_aln_prob = 1.0;
        name = r1.name; as = r1.as; has_as = r1.has_as; if (keep_reads) { sd.as_sum = r1.has_as ? r1.as : 0; sd.put(0, r1); } return true;
      }
      const bool mate_mapped = !(r1.flag & F_MUNMAP);
      if (mapped && mate_mapped && (r1.flag & F_PROPER) && r1.ref == r1.mref) {   // MappedConcordantPair: this record and the next one
        if (!next_record(r2, err)) { if (err.empty()) err = "the SAM file ends in the middle of a read pair (" + r1.name + ")"; return false; }
        if (!(r1.flag & F_PAIRED) || !(r2.flag & F_PAIRED)) { err = "found an unpaired read in a paired-end library; the two ends of a pair must be adjacent (" + r1.name + ")"; return false; }
        if (base_name(r1.name) != base_name(r2.name)) cnt.num_suspicious_pairs++;
        if (keep_reads) {   // AlignmentModel::update / logLikelihood (AlignmentModel.cpp:258-268, :436-441): the record with the smaller position is scored with the left matrices; on a tie the SECOND record of the file is
          sd.as_sum = (r1.has_as ? r1.as : 0) + ((r2.flag & F_UNMAP) || !r2.has_as ? 0 : r2.as);
          const bool first_left = r1.pos < r2.pos; Rec c1 = r1, c2 = r2; sd.put(first_left ? 0 : 1, c1); sd.put(first_left ? 1 : 0, c2); }
        if (r1.flag & F_READ2) std::swap(r1, r2);
        memset(&a, 0, sizeof(a)); a.tid = (uint32_t)r1.ref; a.pos = r1.pos; a.mate_pos = r2.pos; a.fwd = !(r1.flag & F_REVERSE); a.mate_fwd = !(r2.flag & F_REVERSE);
        a.read_len = (uint16_t)std::min<uint32_t>(r1.len, 65535u); a.mate_len = (uint16_t)std::min<uint32_t>(r2.len, 65535u);
        a.frag_len = (uint32_t)std::abs(r1.pos - r2.pos) + (r1.pos < r2.pos ? r2.len : r1.len);   // ReadPair::fragLen
        a.mate_status = SQ_MS_PAIRED_END_PAIRED; a.format_id = hit_type_pair(r1.pos, a.fwd, r2.pos, a.mate_fwd);
        a.score = r1.as; a.mate_score = (r2.flag & F_UNMAP) ? 0 : r2.as; a.est_aln_prob = 1.0;
        name = base_name(r1.name); as = a.score + a.mate_score; has_as = r1.has_as; return true;   // ReadPair::getAS: the sum of the mapped ends' AS tags
      }
      if (mapped) {   // MappedOrphan (also: not a proper pair, or the ends on different targets — BAMQueue.tpp:308-331)
        memset(&a, 0, sizeof(a)); a.tid = (uint32_t)r1.ref; a.pos = r1.pos; a.fwd = !(r1.flag & F_REVERSE); a.read_len = (uint16_t)std::min<uint32_t>(r1.len, 65535u);
        a.mate_status = (r1.flag & F_READ1) ? SQ_MS_PAIRED_END_LEFT : SQ_MS_PAIRED_END_RIGHT; a.format_id = hit_type_single(a.fwd); a.score = r1.as; a.est_aln_prob = 1.0;
        name = base_name(r1.name); as = r1.as; has_as = r1.has_as; if (keep_reads) { sd.as_sum = r1.has_as ? r1.as : 0; sd.put((r1.flag & F_READ1) ? 0 : 1, r1); } return true;
      }
      if (mate_mapped) continue;   // UnmappedOrphan: its mate's record carries the alignment
      // UnmappedPair: both records of the pair are consumed
      if (!next_record(r2, err)) return false;
      cnt.num_unaligned++;
    }
  }
};

extern "C" int sq_sam_open(const char* path, int paired_library, sq_sam** out) {
  if (!path || !out) { sq_set_error("sq_sam_open: bad arguments"); return SQ_ERR_ARG; }
  gzFile f = gzopen(path, "rb"); if (!f) { sq_set_error("cannot open alignment file '%s'", path); return SQ_ERR_IO; }
  gzbuffer(f, 1 << 20);
  sq_sam* s = new sq_sam(); s->f = f; s->path = path; s->paired = paired_library != 0; s->buf.resize(4 << 20);
  { // BGZF?  (the first member names its compressed size in a 'BC' extra field)
    FILE* rf = fopen(path, "rb"); unsigned char h[64]; const size_t got = rf ? fread(h, 1, sizeof h, rf) : 0; if (rf) fclose(rf);
    if (got >= 28 && sqio::BgzfSource::member_size(h, got) && !(getenv("SQ_SAM_BGZF") && atoi(getenv("SQ_SAM_BGZF")) == 0)) {
      int fd = open(path, O_RDONLY); struct stat sb;
      if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size >= 28) {
        void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
          const unsigned nt = std::min(16u, std::max(2u, std::thread::hardware_concurrency() / 4));
          s->pool.reset(new sqio::Pool(nt)); s->bg.reset(new sqio::BgzfSource());
          s->bg->map = std::make_shared<sqio::Mapping>(); s->bg->map->p = m; s->bg->map->n = (size_t)sb.st_size; s->bg->base = (const uint8_t*)m; s->bg->n = (size_t)sb.st_size;
          s->bg->pool = s->pool.get(); s->bg->path = path; s->bg->window = 2 * nt; (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
        }
      }
      if (fd >= 0) close(fd);
    }
  }
  // the header: @SQ lines name the targets in the order the records refer to them
  std::string l; bool first = true;
  for (;;) {
    // peek: header lines start with '@'; the first record line is kept for next_record
    if (s->bpos == s->bend) { const int n = s->fill(s->buf.data(), s->buf.size()); if (n <= 0) { s->eof = true; break; } s->bpos = 0; s->bend = (size_t)n; }
    if (first) { first = false;
      if (s->bend - s->bpos >= 4 && !memcmp(s->buf.data() + s->bpos, "BAM\1", 4)) {   // [r4] BAM: magic, l_text, text, n_ref, then (l_name, name, l_ref) per target
        s->bam = true; uint8_t b4[4]; bool ok = s->read_exact(b4, 4) && s->read_exact(b4, 4);
        if (ok) { int32_t lt = sq_sam::i32(b4); ok = lt >= 0; std::vector<char> skip; std::string tail; while (ok && lt > 0) { skip.resize((size_t)std::min(lt, 1 << 20)); ok = s->read_exact(skip.data(), skip.size()); lt -= (int32_t)skip.size();
            if (ok) { tail.append(skip.data(), skip.size()); size_t at = 0; while ((at = tail.find("@PG", at)) != std::string::npos) { const size_t e = tail.find('\n', at); if (tail.substr(at, e == std::string::npos ? std::string::npos : e - at).find("ID:bowtie2") != std::string::npos) s->bowtie2 = true; at += 3; } if (tail.size() > 4096) tail.erase(0, tail.size() - 4096); } } }
        int32_t nref = 0; if (ok) { ok = s->read_exact(b4, 4); nref = sq_sam::i32(b4); ok = ok && nref >= 0; }
        for (int32_t i = 0; ok && i < nref; ++i) {
          ok = s->read_exact(b4, 4); const int32_t ln = sq_sam::i32(b4); if (!ok || ln <= 0 || ln > (1 << 20)) { ok = false; break; }
          std::string nm((size_t)ln, '\0'); ok = s->read_exact(&nm[0], (size_t)ln) && s->read_exact(b4, 4); if (!ok) break;
          nm.resize(strnlen(nm.c_str(), nm.size())); s->by_name[nm] = (int32_t)s->names.size(); s->names.push_back(nm); s->lens.push_back((uint32_t)sq_sam::i32(b4));
        }
        if (!ok) { gzclose(f); delete s; sq_set_error("'%s': damaged BAM header", path); return SQ_ERR_IO; }
        break;
      } }
    if (s->buf[s->bpos] != '@') break;
    if (!s->getline(l)) break;
    if (!strncmp(l.c_str(), "@PG", 3) && l.find("ID:bowtie2") != std::string::npos) s->bowtie2 = true;   // the aligner whose AS tags weigh the error model's updates (SalmonQuantifyAlignments.cpp:265-285)
    if (!strncmp(l.c_str(), "@SQ", 3)) {
      std::string sn; uint32_t ln = 0; size_t p = 0;
      while ((p = l.find('\t', p)) != std::string::npos) { ++p; if (!l.compare(p, 3, "SN:")) { const size_t e = l.find('\t', p); sn = l.substr(p + 3, e == std::string::npos ? std::string::npos : e - p - 3); } else if (!l.compare(p, 3, "LN:")) ln = (uint32_t)strtoul(l.c_str() + p + 3, nullptr, 10); }
      if (!sn.empty()) { s->by_name[sn] = (int32_t)s->names.size(); s->names.push_back(sn); s->lens.push_back(ln); }
    }
  }
  if (s->names.empty()) { const bool was_bam = s->bam; gzclose(f); delete s; sq_set_error(was_bam ? "'%s' names no reference sequences in its BAM header" : "'%s' has no @SQ header lines: the targets of the alignments are unknown", path); return SQ_ERR_IO; }
  s->tid_map.resize(s->names.size()); for (size_t i = 0; i < s->names.size(); ++i) s->tid_map[i] = (uint32_t)i;
  *out = s; return SQ_OK;
}
// [r5] the FLAG field of the first record (SAM text, gzip, BAM alike): what `-l A` looks at to tell a paired from a single-end file
extern "C" int sq_sam_first_flag(const char* path, int* flag) {
  if (!flag) { sq_set_error("sq_sam_first_flag: bad arguments"); return SQ_ERR_ARG; }
  sq_sam* s = nullptr; const int rc = sq_sam_open(path, 0, &s); if (rc) return rc;
  Rec r; std::string err; const bool got = s->next_record(r, err);
  if (!got) { if (err.empty()) err = s->io_err.empty() ? "no alignment records" : s->io_err; sq_set_error("%s: %s", path, err.c_str()); sq_sam_close(s); return SQ_ERR_IO; }
  *flag = r.flag; sq_sam_close(s); return SQ_OK;
}
extern "C" int sq_sam_keep_reads(sq_sam* s, int on) { if (!s) { sq_set_error("sq_sam_keep_reads: bad arguments"); return SQ_ERR_ARG; } s->keep_reads = on != 0; return SQ_OK; }
extern "C" int sq_sam_reads(sq_sam* s, sq_aln_reads* out) {
  if (!s || !out || !s->keep_reads) { sq_set_error("sq_sam_reads: the reader does not keep the reads (sq_sam_keep_reads)"); return SQ_ERR_STATE; }
  out->num_alignments = s->alns.size(); out->cig_off = s->r_cig_off.data(); out->cigar = s->r_cig.data(); out->seq_off = s->r_seq_off.data(); out->seq = s->r_seq.data();
  out->pos = s->r_pos.data(); out->aligner_score = s->r_score.data(); return SQ_OK;
}
extern "C" uint32_t sq_sam_num_refs(const sq_sam* s) { return s ? (uint32_t)s->names.size() : 0; }
extern "C" const char* sq_sam_ref_name(const sq_sam* s, uint32_t i) { return (s && i < s->names.size()) ? s->names[i].c_str() : ""; }
extern "C" uint32_t sq_sam_ref_len(const sq_sam* s, uint32_t i) { return (s && i < s->lens.size()) ? s->lens[i] : 0; }
extern "C" int sq_sam_set_tid_map(sq_sam* s, const uint32_t* map, uint32_t n) {
  if (!s || !map || n != s->names.size()) { sq_set_error("sq_sam_set_tid_map: one entry per @SQ line"); return SQ_ERR_ARG; }
  s->tid_map.assign(map, map + n); return SQ_OK;
}
extern "C" void sq_sam_close(sq_sam* s) { if (s) { s->cur = PgzBuf(); s->bg.reset(); s->pool.reset(); if (s->f) gzclose(s->f); delete s; } }

extern "C" int sq_sam_next(sq_sam* s, uint32_t max_frags, int use_as_scores, double score_exp, sq_aln_batch* out, sq_sam_counts* counts) {
  if (!s || !out || !max_frags) { sq_set_error("sq_sam_next: bad arguments"); return SQ_ERR_ARG; }
  s->off.assign(1, 0); s->alns.clear(); s->as_of.clear(); s->has_as_of.clear(); s->sides.clear();
  std::string err; uint32_t nfrag = 0;
  auto close_fragment = [&](size_t a0) {   // order by transcript (AlignmentGroup::sortHits), then the AS-based conditional probabilities
    const size_t a1 = s->alns.size(); if (a1 == a0) return;
    std::vector<uint32_t> ord(a1 - a0); for (size_t i = 0; i < ord.size(); ++i) ord[i] = (uint32_t)i;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return s->alns[a0 + x].tid < s->alns[a0 + y].tid; });
    std::vector<sq_aln> tmp(ord.size()); std::vector<int32_t> tas(ord.size()); bool all_as = true;
    if (s->keep_reads) {   // the side records follow their alignments; a record without its sequence (a secondary alignment's "*") borrows the fragment's stored one of the same read, turned round if the strands differ
      std::vector<Side> ts(ord.size()); for (size_t i = 0; i < ord.size(); ++i) ts[i] = std::move(s->sides[a0 + ord[i]]);
      for (size_t i = 0; i < ord.size(); ++i) for (int k = 0; k < 2; ++k) { Side& x = ts[i]; if (x.cig[k].empty() || !x.seq[k].empty()) continue;
        for (size_t j = 0; j < ord.size() && x.seq[k].empty(); ++j) for (int q = 0; q < 2; ++q) if (!ts[j].seq[q].empty() && ts[j].mate[q] == x.mate[k]) { x.seq[k] = ts[j].seq[q];
          if (ts[j].rev[q] != x.rev[k]) { std::reverse(x.seq[k].begin(), x.seq[k].end()); for (auto& b : x.seq[k]) b = (uint8_t)(3 - b); } break; } }
      for (size_t i = 0; i < ord.size(); ++i) s->sides[a0 + i] = std::move(ts[i]);
    }
    for (size_t i = 0; i < ord.size(); ++i) { tmp[i] = s->alns[a0 + ord[i]]; tas[i] = s->as_of[a0 + ord[i]]; all_as = all_as && s->has_as_of[a0 + ord[i]]; }
    int32_t best = INT32_MIN; for (int32_t v : tas) best = std::max(best, v);
    for (size_t i = 0; i < ord.size(); ++i) { if (use_as_scores && all_as) tmp[i].est_aln_prob = std::exp(-score_exp * (double)(best - tas[i])); s->alns[a0 + i] = tmp[i]; }
    if (use_as_scores && !all_as) s->cnt.num_frags_without_as++;
    s->off.push_back(a1); ++nfrag; s->cnt.num_fragments++;
  };
  size_t frag_start = 0; std::string cur_name; bool open = false;
  for (;;) {
    sq_aln a; std::string name; int32_t as = 0; bool has_as = false; Side sd;
    if (s->have_next) { a = s->next_aln; name = s->next_name; as = s->next_as; has_as = s->next_has_as; sd = std::move(s->next_side); s->have_next = false; }
    else if (!s->next_alignment(a, name, as, has_as, err, sd)) {
      if (err.empty() && !s->io_err.empty()) err = s->io_err;
      if (!err.empty()) { sq_set_error("%s: %s", s->path.c_str(), err.c_str()); return SQ_ERR_IO; } break; }
    if (open && name != cur_name) {
      close_fragment(frag_start); open = false;
      if (nfrag == max_frags) { s->have_next = true; s->next_aln = a; s->next_name = name; s->next_as = as; s->next_has_as = has_as; s->next_side = std::move(sd); break; }
    }
    if (!open) { open = true; cur_name = name; frag_start = s->alns.size(); }
    const uint32_t t = a.tid < s->tid_map.size() ? s->tid_map[a.tid] : 0xFFFFFFFFu;
    if (t == 0xFFFFFFFFu) { s->cnt.num_skipped_unknown_target++; continue; }
    a.tid = t; s->alns.push_back(a); s->as_of.push_back(as); s->has_as_of.push_back(has_as ? 1 : 0); s->cnt.num_alignments++;
    if (s->keep_reads) s->sides.push_back(std::move(sd));
  }
  if (open && !s->have_next) close_fragment(frag_start);
  if (s->keep_reads) {   // flat arrays for sq_sam_reads
    const size_t na = s->alns.size(); s->r_cig_off.assign(1, 0); s->r_seq_off.assign(1, 0); s->r_cig.clear(); s->r_seq.clear(); s->r_pos.resize(2 * na); s->r_score.resize(na);
    for (size_t i = 0; i < na; ++i) { const Side& x = s->sides[i]; s->r_score[i] = s->bowtie2 ? x.as_sum : 0;
      for (int k = 0; k < 2; ++k) { s->r_cig.insert(s->r_cig.end(), x.cig[k].begin(), x.cig[k].end()); s->r_seq.insert(s->r_seq.end(), x.seq[k].begin(), x.seq[k].end()); s->r_cig_off.push_back(s->r_cig.size()); s->r_seq_off.push_back(s->r_seq.size()); s->r_pos[2 * i + k] = x.pos[k]; } }
  }
  out->n = nfrag; out->read_off = s->off.data(); out->aln = s->alns.data(); out->aln_cap = s->alns.size(); out->map_type = nullptr;
  if (counts) *counts = s->cnt;
  return SQ_OK;
}
