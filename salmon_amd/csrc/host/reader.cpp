// host/reader.cpp — the host read pipeline in front of seam B1.
//
// Replaces the reference's FQFeeder parser threads + per-worker chunk queues
// (reference src/quant/SalmonQuantify.cpp:2419-2443, include/salmon/internal/io/FastxReader.hpp):
// one producer thread per mate stream inflates (zlib; plain text goes through the same path) and
// splits FASTQ/FASTA records into blocks of sequences; sq_reader_next() interleaves the two
// streams into one read batch — concatenated bases + offsets, the exact layout sq_map_batch /
// sq_map_submit take — inside one of a few rotating, page-locked host buffers, so the caller can
// keep one batch on every mapping lane (H2D + mapping) while the next one is being assembled.
// A single gzip stream cannot be inflated in parallel; several files per mate are consumed in
// order by the same producer, as the reference does.
//
// Fast path (round 2): FASTQ with one-line sequences and qualities — what sequencers write.  A stream thread cuts the input into
// ~8 MB chunks at record boundaries (plain files are mmap'ed: the chunk is a window of the page cache, nothing is copied; .gz files
// are inflated by the stream thread — the sequential floor of a gzip stream — into chunk buffers; BGZF files, whose 64 KB members are
// independent, are inflated by the pool), a pool of workers finds the
// records of the chunks in parallel (memchr per line, every record checked for the 4-line shape), and sq_reader_next assembles the
// interleaved batch in the page-locked slot with the same pool (per-part byte totals, a scan, parallel copies): one copy per base from
// the page cache to the buffer the GPU reads.  FASTA, multi-line records and SQ_READER_SAFE=1 take the kseq-rules path below.
#include "index.h"
#include "reader_dev.h"
#include "crc32_fast.h"
#include "bgzf_source.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <functional>
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include "pgzip.h"
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
using sqio::Pool; using sqio::Mapping; using sqio::BgzfSource;

struct RecBlock { std::vector<char> seq; std::vector<uint32_t> len; std::vector<char> names; std::vector<uint32_t> nlen; };   // sequences back to back (+ names on request)

// read name = the header up to the first blank (kseq's name field: what the reference prints in unmapped_names.txt; the SAM writer
// drops a trailing /1 or /2 itself)
inline size_t name_len(const char* h, size_t n) {
  size_t l = 0; while (l < n && h[l] != ' ' && h[l] != '\t' && h[l] != '\r') ++l;
  return l;
}

// bounded single-producer / single-consumer queue of record blocks
struct BlockQueue {
  std::mutex mu;
  std::condition_variable cv_put, cv_get;
  std::deque<std::unique_ptr<RecBlock>> q;
  bool done = false;
  std::string err;
  size_t cap = 8;
  void put(std::unique_ptr<RecBlock> b) {
    std::unique_lock<std::mutex> lk(mu);
    cv_put.wait(lk, [&] { return q.size() < cap || done; });
    if (done) return;
    q.push_back(std::move(b));
    cv_get.notify_one();
  }
  void finish(const std::string& e = std::string()) {
    {
      std::lock_guard<std::mutex> lk(mu);
      done = true;
      if (!e.empty()) err = e;
    }
    cv_get.notify_all();
    cv_put.notify_all();
  }
  std::unique_ptr<RecBlock> get() {
    std::unique_lock<std::mutex> lk(mu);
    cv_get.wait(lk, [&] { return !q.empty() || done; });
    if (q.empty()) return nullptr;
    auto b = std::move(q.front());
    q.pop_front();
    cv_put.notify_one();
    return b;
  }
};

// one mate stream: inflate + split records.  FASTQ and FASTA, sequences and qualities on one or several lines
// (what the reference's kseq-based parser accepts), LF or CRLF, blank lines between records tolerated.
void produce(std::vector<std::string> files, BlockQueue* out, bool keep_names) {
  const size_t BUF = 4u << 20; std::vector<char> buf(BUF); const uint32_t PER_BLOCK = 16384;
  auto fresh = [&] { auto b = std::make_unique<RecBlock>(); b->seq.reserve((size_t)PER_BLOCK * 128); b->len.reserve(PER_BLOCK); return b; };
  auto blk = fresh();
  for (const auto& path : files) {
    gzFile f = gzopen(path.c_str(), "rb"); if (!f) { out->finish("cannot open '" + path + "'"); return; }
    gzbuffer(f, 1 << 20);
    uint64_t have = 0; int state = 0; bool fastq = true; std::string bad;   // state: 0 header, 1 sequence lines, 3 quality lines
    uint32_t cur_len = 0; uint64_t qual_len = 0;                              // bases of the open record / quality characters seen for it
    auto close_record = [&] {
      blk->len.push_back(cur_len); ++have; cur_len = 0; qual_len = 0; state = 0;
      if (blk->len.size() == PER_BLOCK) { out->put(std::move(blk)); blk = fresh(); }
    };
    // kseq-style records: the sequence (and the quality string) may span several lines; a FASTA record ends at the
    // next '>' line, a FASTQ sequence at the '+' line, a FASTQ record once the quality is as long as the sequence
    // (so a quality line may begin with '@')
    auto line = [&](const char* ls, size_t ll) {
      if (ll && ls[ll - 1] == '\r') --ll;
      if (state == 1 && !fastq && ll && ls[0] == '>') close_record();       // falls through to the header branch
      if (state == 0) {
        if (!ll) return;                                   // blank line between records
        if (ls[0] == '@') fastq = true; else if (ls[0] == '>') fastq = false;
        else { bad = "'" + path + "': record " + std::to_string(have + 1) + " does not start with '@' or '>'"; return; }
        state = 1; cur_len = 0; qual_len = 0;
        if (keep_names) { const size_t nl = name_len(ls + 1, ll - 1); blk->names.insert(blk->names.end(), ls + 1, ls + 1 + nl); blk->nlen.push_back((uint32_t)nl); }
      } else if (state == 1) {
        if (fastq && ll && ls[0] == '+') { state = 3; if (cur_len == 0) close_record(); return; }
        blk->seq.insert(blk->seq.end(), ls, ls + ll); cur_len += (uint32_t)ll;
      } else {                                             // quality
        qual_len += ll;
        if (qual_len > cur_len) {
          bad = "'" + path + "': record " + std::to_string(have + 1) + " has a quality string longer than its sequence";
          return;
        }
        if (qual_len == cur_len) close_record();
      }
    };
    std::vector<char> pend; size_t start = 0;
    for (;;) {
      const int n = gzread(f, buf.data(), (unsigned)BUF);
      if (n < 0) { int e; std::string m = gzerror(f, &e); gzclose(f); out->finish("read error in '" + path + "': " + m); return; }
      if (n > 0) {
        if (start == pend.size()) { pend.clear(); start = 0; }
        else if (start > (1u << 20)) { pend.erase(pend.begin(), pend.begin() + (ptrdiff_t)start); start = 0; }
        pend.insert(pend.end(), buf.data(), buf.data() + n);
        for (;;) {
          const char* s0 = pend.data() + start; const char* nl = (const char*)memchr(s0, '\n', pend.size() - start);
          if (!nl) break;
          line(s0, (size_t)(nl - s0)); start = (size_t)(nl - pend.data()) + 1;
          if (!bad.empty()) { gzclose(f); out->finish(bad); return; }
        }
      } else {
        if (start < pend.size()) {
          line(pend.data() + start, pend.size() - start);
          if (!bad.empty()) {
            gzclose(f);
            out->finish(bad);
            return;
          }
        }
        break;
      }
    }
    gzclose(f);
    if (state == 1 && !fastq) close_record();              // the last FASTA record ends with the file
    if (state != 0) { out->finish("'" + path + "': truncated record at end of file (after " + std::to_string(have) + " records)"); return; }
  }
  if (!blk->len.empty()) out->put(std::move(blk));
  out->finish();
}


// ---- fast path ----------------------------------------------------------------------------------------------------------------
struct Chunk {   // a run of whole records of one stream, in stream order
  const char* text = nullptr; size_t bytes = 0; std::unique_ptr<char[]> own; std::shared_ptr<Mapping> map; std::shared_ptr<void> hold;   // window of an mmap, an owned buffer (.gz), or a window of somebody's buffer kept alive by `hold`
  std::vector<uint32_t> pos, len;   // per record: where its bases start in `text`, how many
  std::vector<uint32_t> npos, nlen; bool keep_names = false;   // on request: where its name starts, how long it is
  std::string path; uint64_t first_record = 0;
  bool done = false; std::string err;
};
struct ChunkQueue {   // ordered, bounded: the stream thread appends chunks, workers complete them, the consumer takes them in order
  std::mutex mu; std::condition_variable cv_done, cv_room; std::deque<std::shared_ptr<Chunk>> q; bool eof = false; std::string err; size_t cap = 24;
  void push(std::shared_ptr<Chunk> c) { std::unique_lock<std::mutex> lk(mu); cv_room.wait(lk, [&] { return q.size() < cap || eof; }); if (eof) return; q.push_back(std::move(c)); }
  void complete(Chunk* c) { { std::lock_guard<std::mutex> lk(mu); c->done = true; } cv_done.notify_all(); }
  void finish(const std::string& e = std::string()) { { std::lock_guard<std::mutex> lk(mu); eof = true; if (!e.empty() && err.empty()) err = e; } cv_done.notify_all(); cv_room.notify_all(); }
  std::shared_ptr<Chunk> pop() {   // next chunk in order once its records are known; nullptr at the end (or on error: see err)
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return (!q.empty() && q.front()->done) || (q.empty() && eof); });
    if (q.empty()) return nullptr;
    auto c = q.front(); q.pop_front(); cv_room.notify_one();
    if (!c->err.empty() && err.empty()) err = c->err;
    return c;
  }
};

// start of the first whole 4-line record at or after `from` (which must be a line start), or `end` if none begins there
const char* record_start(const char* from, const char* end) {
  const char* p = from;
  while (p < end) {
    const char* e0 = (const char*)memchr(p, '\n', (size_t)(end - p)); if (!e0) return end;
    if (*p == '@') {
      const char* l1 = e0 + 1; const char* e1 = l1 < end ? (const char*)memchr(l1, '\n', (size_t)(end - l1)) : nullptr;
      if (!e1) return end;
      const char* l2 = e1 + 1; const char* e2 = l2 < end ? (const char*)memchr(l2, '\n', (size_t)(end - l2)) : nullptr;
      if (!e2) return end;
      const char* l3 = e2 + 1; const char* e3 = l3 < end ? (const char*)memchr(l3, '\n', (size_t)(end - l3)) : end;
      if (!e3) e3 = end;
      if (*l2 == '+' && (e3 - l3) - ((e3 > l3 && e3[-1] == '\r') ? 1 : 0) == (e1 - l1) - ((e1 > l1 && e1[-1] == '\r') ? 1 : 0)) return p;
    }
    p = e0 + 1;
  }
  return end;
}

void parse_chunk(Chunk* c) {
  const char* p = c->text; const char* end = c->text + c->bytes;
  c->pos.reserve(c->bytes / 200 + 16); c->len.reserve(c->bytes / 200 + 16);
  uint64_t rec = c->first_record;
  auto fail = [&](const char* what) { c->err = "'" + c->path + "': record " + std::to_string(rec + 1) + " " + what + " (multi-line FASTQ? set SQ_READER_SAFE=1)"; };
  while (p < end) {
    if (*p == '\n' || *p == '\r') { ++p; continue; }                       // blank line between records
    if (*p != '@') { fail("does not start with '@'"); return; }
    const char* e0 = (const char*)memchr(p, '\n', (size_t)(end - p)); if (!e0) { fail("is truncated"); return; }
    const char* l1 = e0 + 1; const char* e1 = (const char*)memchr(l1, '\n', (size_t)(end - l1)); if (!e1) { fail("is truncated"); return; }
    const char* l2 = e1 + 1; const char* e2 = l2 < end ? (const char*)memchr(l2, '\n', (size_t)(end - l2)) : nullptr; if (!e2 || *l2 != '+') { fail(e2 ? "has no '+' line after one sequence line" : "is truncated"); return; }
    const char* l3 = e2 + 1; const char* e3 = l3 <= end ? (const char*)memchr(l3, '\n', (size_t)(end - l3)) : nullptr; if (!e3) e3 = end;
    size_t sl = (size_t)(e1 - l1); if (sl && l1[sl - 1] == '\r') --sl;
    size_t ql = (size_t)(e3 - l3); if (ql && l3[ql - 1] == '\r') --ql;
    if (ql != sl) { fail(ql > sl ? "has a quality string longer than its sequence" : "has a quality string shorter than its sequence: truncated record"); return; }
    c->pos.push_back((uint32_t)(l1 - c->text)); c->len.push_back((uint32_t)sl); ++rec;
    if (c->keep_names) { c->npos.push_back((uint32_t)(p + 1 - c->text)); c->nlen.push_back((uint32_t)name_len(p + 1, (size_t)(e0 - p - 1))); }
    p = e3 < end ? e3 + 1 : end;
  }
}

// does the file look like 4-line FASTQ?  Every record that lies wholly inside the first bytes of the file must have the shape
// '@' header / one sequence line / '+' line / one quality line of the same length (wrapped files wrap from the first record on)
bool looks_like_simple_fastq(const char* b, size_t n) {
  if (!n || b[0] != '@') return false;
  const char* end = b + n; const char* p = b; int seen = 0;
  while (p < end) {
    if (*p == '\n' || *p == '\r') { ++p; continue; }
    const char* e0 = (const char*)memchr(p, '\n', (size_t)(end - p)); if (!e0) break;
    const char* l1 = e0 + 1; const char* e1 = l1 < end ? (const char*)memchr(l1, '\n', (size_t)(end - l1)) : nullptr; if (!e1) break;
    const char* l2 = e1 + 1; const char* e2 = l2 < end ? (const char*)memchr(l2, '\n', (size_t)(end - l2)) : nullptr; if (!e2) break;
    const char* l3 = e2 + 1; const char* e3 = l3 < end ? (const char*)memchr(l3, '\n', (size_t)(end - l3)) : nullptr; if (!e3) break;
    size_t sl = (size_t)(e1 - l1); if (sl && l1[sl - 1] == '\r') --sl;
    size_t ql = (size_t)(e3 - l3); if (ql && l3[ql - 1] == '\r') --ql;
    if (*p != '@' || *l2 != '+' || sl != ql) return false;
    ++seen; p = e3 + 1;
  }
  return seen > 0 || n < 65536;   // a tiny file without one whole record: let the parser say what is wrong with it
}

const size_t CHUNK_BYTES = 8u << 20;

// one mate stream on the fast path; returns false (nothing consumed) if the first file is not simple FASTQ and the safe path should run
void produce_fast(std::vector<std::string> files, ChunkQueue* out, Pool* pool, bool keep_names) {
  uint64_t nrec_before = 0;
  auto dispatch = [&](std::shared_ptr<Chunk> c) {
    c->keep_names = keep_names;
    out->push(c);
    pool->submit([c, out] { parse_chunk(c.get()); out->complete(c.get()); });
  };
  for (const auto& path : files) {
    const bool gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    if (!gz) {
      int fd = open(path.c_str(), O_RDONLY); if (fd < 0) { out->finish("cannot open '" + path + "'"); return; }
      struct stat sb; if (fstat(fd, &sb) != 0) { close(fd); out->finish("cannot stat '" + path + "'"); return; }
      const size_t n = (size_t)sb.st_size;
      if (n == 0) { close(fd); continue; }
      void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); close(fd);
      if (m == MAP_FAILED) { out->finish("cannot map '" + path + "'"); return; }
      (void)madvise(m, n, MADV_SEQUENTIAL);
      auto map = std::make_shared<Mapping>(); map->p = m; map->n = n;
      const char* base = (const char*)m; const char* end = base + n; const char* p = base;
      while (p < end) {
        const char* q = p + CHUNK_BYTES;
        if (q >= end) q = end;
        else { const char* nl = (const char*)memchr(q, '\n', (size_t)(end - q)); q = nl ? record_start(nl + 1, end) : end; }
        auto c = std::make_shared<Chunk>(); c->text = p; c->bytes = (size_t)(q - p); c->map = map; c->path = path; c->first_record = nrec_before;
        nrec_before += c->bytes / 250;   // only used in messages
        dispatch(c); p = q;
      }
    } else {
      // BGZF: the members are inflated by the pool; any other gzip file by this thread through zlib
      std::unique_ptr<BgzfSource> bg;
      { int fd = open(path.c_str(), O_RDONLY); struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && sb.st_size >= 28) {
          void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
          if (m != MAP_FAILED) {
            if (BgzfSource::member_size((const uint8_t*)m, (size_t)sb.st_size)) {
              bg.reset(new BgzfSource()); bg->map = std::make_shared<Mapping>(); bg->map->p = m; bg->map->n = (size_t)sb.st_size;
              bg->base = (const uint8_t*)m; bg->n = (size_t)sb.st_size; bg->pool = pool; bg->path = path; bg->window = std::max<size_t>(8, 2 * pool->th.size());
              (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
            } else munmap(m, (size_t)sb.st_size);
          }
        }
        if (fd >= 0) close(fd); }
      // [r3] any other gzip file of some size: cut into pieces that the pool inflates in parallel (pgzip.h); smaller files stay on one zlib stream
      std::shared_ptr<Mapping> pmap; PgzStream* pz = nullptr;
      struct PzGuard { PgzStream*& p; ~PzGuard() { if (p) pgz_close(p); } } pzg{pz};
      if (!bg) {
        int fd = open(path.c_str(), O_RDONLY); struct stat sb;
        const long min_bytes = getenv("SQ_READER_PGZ_MIN") ? atol(getenv("SQ_READER_PGZ_MIN")) : (8L << 20);   // smaller files: one zlib stream is as fast (the variable: tests reach the pieces with small files)
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size >= min_bytes) {
          void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
          if (m != MAP_FAILED) {
            pmap = std::make_shared<Mapping>(); pmap->p = m; pmap->n = (size_t)sb.st_size; (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
            const unsigned th = (unsigned)std::max<size_t>(1, std::min<size_t>(pool->th.size(), 16));
            const size_t piece = getenv("SQ_READER_PGZ_PIECE") ? (size_t)atol(getenv("SQ_READER_PGZ_PIECE")) : std::max<size_t>(1u << 20, std::min<size_t>(4u << 20, (size_t)sb.st_size / (4 * th)));
            pz = pgz_open((const uint8_t*)m, (size_t)sb.st_size, [pool](std::function<void()> f) { pool->submit(std::move(f)); }, th, piece);
            if (!pz) pmap.reset();
          }
        }
        if (fd >= 0) close(fd);
      }
      gzFile f = nullptr;
      if (!bg && !pz) { f = gzopen(path.c_str(), "rb"); if (!f) { out->finish("cannot open '" + path + "'"); return; } gzbuffer(f, 1 << 20); }
      std::vector<char> carry;
      if (pz || bg) {   // [r3] the pieces' text (gzip by pieces; [r4] BGZF member groups) is parsed where it lies: only the record that straddles two buffers is copied
        auto last_whole = [](const char* b, const char* e) -> const char* {   // the last record start in [b, e) whose four lines are all there (b if none)
          size_t back = std::min<size_t>((size_t)(e - b), 1u << 16);
          for (;;) {
            const char* from = e - back; const char* nl = (const char*)memchr(from, '\n', (size_t)(e - from));
            const char* last = nullptr; const char* x = (from == b) ? b : (nl ? nl + 1 : e);
            while (x < e) { const char* r0 = record_start(x, e); if (r0 >= e) break; last = r0; const char* n0 = (const char*)memchr(r0, '\n', (size_t)(e - r0)); x = n0 ? n0 + 1 : e; }
            if (last) return last;
            if (back == (size_t)(e - b)) return b;
            back = std::min<size_t>((size_t)(e - b), back * 4);
          }
        };
        auto owned = [&](const char* a, size_t na, const char* b2, size_t nb) {
          auto c = std::make_shared<Chunk>(); c->own.reset(new char[na + nb + 1]); memcpy(c->own.get(), a, na); if (nb) memcpy(c->own.get() + na, b2, nb);
          c->text = c->own.get(); c->bytes = na + nb; c->path = path; c->first_record = nrec_before; nrec_before += (na + nb) / 250; dispatch(c);
        };
        for (;;) {
          PgzBuf B; std::string e; const int rc = pz ? pgz_next(pz, &B, &e) : bg->next_buf(&B);
          if (rc < 0) { out->finish(pz ? "'" + path + "': " + e : bg->err); return; }
          if (rc == 0) break;
          const char* b = B.p; const char* end = b + B.n; const char* p = b;
          if (!carry.empty()) {   // the record begun in the previous buffer: its missing lines are at the head of this one
            size_t have_nl = 0; for (char ch : carry) have_nl += ch == '\n';
            const size_t lines = have_nl + (carry.back() != '\n' ? 1 : 0), want_nl = (lines + 3) / 4 * 4;   // the carry holds whole records and the head of one more
            while (have_nl < want_nl && p < end) { const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p)); if (!nl) { p = end; break; } p = nl + 1; ++have_nl; }
            if (have_nl < want_nl) { carry.insert(carry.end(), b, end); continue; }   // a buffer shorter than a record
            owned(carry.data(), carry.size(), b, (size_t)(p - b)); carry.clear();
          }
          const char* cut = last_whole(p, end);
          // the last record of [p, end) may be complete only by luck of where the buffer ends: it waits for the next buffer (or the end of the file)
          while (p < cut) {
            const char* q = p + CHUNK_BYTES;
            if (q >= cut) q = cut; else { const char* nl = (const char*)memchr(q, '\n', (size_t)(cut - q)); q = nl ? record_start(nl + 1, cut) : cut; }
            auto c = std::make_shared<Chunk>(); c->text = p; c->bytes = (size_t)(q - p); c->hold = B.hold; c->path = path; c->first_record = nrec_before; nrec_before += c->bytes / 250;
            dispatch(c); p = q;
          }
          carry.assign(cut, end);
        }
        if (!carry.empty()) owned(carry.data(), carry.size(), nullptr, 0);
        continue;   // next file
      }
      for (;;) {
        auto c = std::make_shared<Chunk>(); c->own.reset(new char[carry.size() + CHUNK_BYTES]);   // not zero-filled: every byte used is written below
        if (!carry.empty()) memcpy(c->own.get(), carry.data(), carry.size());
        size_t have = carry.size(); carry.clear();
        size_t got = 0; bool eof = false;
        while (got < CHUNK_BYTES) {
          long r;
          if (pz) { std::string e; r = pgz_read(pz, c->own.get() + have + got, CHUNK_BYTES - got, &e); if (r < 0) { out->finish("'" + path + "': " + e); return; } }
          else {
            r = gzread(f, c->own.get() + have + got, (unsigned)std::min<size_t>(CHUNK_BYTES - got, 1u << 30));
            if (r < 0) { int e; std::string msg = gzerror(f, &e); gzclose(f); out->finish("read error in '" + path + "': " + msg); return; }
          }
          if (r == 0) { eof = true; break; }
          got += (size_t)r;
        }
        const size_t tot = have + got;
        if (tot == 0) break;
        size_t cut = tot;
        if (!eof) {   // keep the incomplete tail for the next chunk: cut at the last record start whose four lines are all here
          const char* b = c->own.get(); const char* e = b + tot; cut = 0;
          size_t back = std::min<size_t>(tot, 1u << 16);
          for (;;) {
            const char* from = e - back; const char* nl = (const char*)memchr(from, '\n', (size_t)(e - from));
            const char* last = nullptr; const char* x = (from == b) ? b : (nl ? nl + 1 : e);
            while (x < e) { const char* r0 = record_start(x, e); if (r0 >= e) break; last = r0; const char* n0 = (const char*)memchr(r0, '\n', (size_t)(e - r0)); x = n0 ? n0 + 1 : e; }
            if (last) { cut = (size_t)(last - b); break; }
            if (back == tot) break;
            back = std::min<size_t>(tot, back * 4);
          }
          if (cut == 0) { if (f) gzclose(f); out->finish("'" + path + "': no FASTQ record boundary in a " + std::to_string(tot) + "-byte window (set SQ_READER_SAFE=1)"); return; }
          carry.assign(c->own.get() + cut, c->own.get() + tot);
        }
        c->text = c->own.get(); c->bytes = cut; c->path = path; c->first_record = nrec_before; nrec_before += cut / 250;
        dispatch(c);
        if (eof) break;
      }
      if (f) gzclose(f);
    }
  }
  out->finish();
}

struct Slot { uint8_t* seq = nullptr; size_t seq_cap = 0; uint64_t* off = nullptr; size_t off_cap = 0; bool pinned = false,
    off_pinned = false; bool busy = false; std::vector<char> names; std::vector<uint64_t> name_off; };

void* host_alloc(size_t bytes, bool* pinned) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) { *pinned = true; return p; }
  // no device (CPU-side tests): pageable memory works the same, only slower to upload
  (void)hipGetLastError();
  *pinned = false;
  return malloc(bytes);
}
void host_free(void* p, bool pinned) { if (!p) return; if (pinned) (void)hipHostFree(p); else free(p); }

}  // namespace

struct sq_reader {
  sq_dev_reader* dev = nullptr;   // [r4] plain 4-line FASTQ files with a device present: the records are split on the GPU (hip/fastq_dev.hip); everything below is then unused
  bool paired = false, keep_names = false; uint32_t batch = 0; BlockQueue q[2]; std::thread th[2];
  // fast path
  bool fast = false; std::unique_ptr<Pool> pool; ChunkQueue cq[2]; std::shared_ptr<Chunk> fc[2]; size_t fidx[2] = {0, 0}; bool fend[2] = {false, false};
  struct Seg { std::shared_ptr<Chunk> c; size_t first, count; };
  // up to `want` records of stream i, in order, without consuming them
  size_t gather(int i, size_t want, std::vector<Seg>& out) {
    size_t got = 0; std::shared_ptr<Chunk> c = fc[i]; size_t idx = fidx[i];
    out.clear();
    // records of chunks beyond the current one are looked at by popping: popped chunks are kept in `ahead`
    size_t ai = 0;
    while (got < want) {
      if (!c || idx >= c->pos.size()) {
        if (ai < ahead[i].size()) { c = ahead[i][ai++]; idx = 0; continue; }
        if (fend[i]) break;
        auto nc = cq[i].pop();
        if (!nc) { fend[i] = true; break; }
        ahead[i].push_back(nc); continue;
      }
      const size_t take = std::min(want - got, c->pos.size() - idx);
      out.push_back({c, idx, take}); got += take; idx += take;
    }
    return got;
  }
  void consume(int i, size_t n) {   // advance the cursor of stream i by n records
    while (n) {
      if (!fc[i] || fidx[i] >= fc[i]->pos.size()) { fc[i] = ahead[i].front(); ahead[i].erase(ahead[i].begin()); fidx[i] = 0; continue; }
      const size_t take = std::min(n, fc[i]->pos.size() - fidx[i]); fidx[i] += take; n -= take;
    }
    if (fc[i] && fidx[i] >= fc[i]->pos.size() && !ahead[i].empty()) { fc[i] = ahead[i].front(); ahead[i].erase(ahead[i].begin()); fidx[i] = 0; }
  }
  std::vector<std::shared_ptr<Chunk>> ahead[2];
  std::unique_ptr<RecBlock> cur[2]; size_t cur_rec[2] = {0, 0}, cur_byte[2] = {0, 0}, cur_nbyte[2] = {0, 0};
  std::vector<Slot> slots; uint64_t total = 0; bool ended = false;
  // [r4] the slots' page-locked buffers are allocated by a thread of their own from the moment the reader is opened (page-locking costs ~0.2 s per GB — as
  // much as inflating what goes into them), slot 0 first; sq_reader_next waits for the slot it is about to fill
  std::thread prealloc; std::mutex amu; std::condition_variable acv; std::vector<char> alloc_state; size_t alloc_asked = 0; bool alloc_stop = false;   // per slot: 0 pending, 1 done, 2 failed / not made
  void start_prealloc() {
    alloc_state.assign(slots.size(), 0);
    int dev_id = -1; if (hipGetDevice(&dev_id) != hipSuccess) { (void)hipGetLastError(); dev_id = -1; }   // the opener's device: the thread below must not wake device 0 in a process that works on another
    prealloc = std::thread([this, dev_id] {
      if (dev_id >= 0 && hipSetDevice(dev_id) != hipSuccess) (void)hipGetLastError();
      const size_t nrec_max = (size_t)batch * (paired ? 2 : 1);
      for (size_t i = 0; i < slots.size(); ++i) {
        // one slot ahead of what has been asked for: a job of one batch locks two buffers, not all of them
        { std::unique_lock<std::mutex> lk(amu); acv.wait(lk, [&] { return alloc_stop || i <= alloc_asked + 1; }); if (alloc_stop) { for (size_t j = i; j < slots.size(); ++j) alloc_state[j] = 2; acv.notify_all(); return; } }
        Slot& S = slots[i]; bool pin = false, ok = true;
        S.off = (uint64_t*)host_alloc((nrec_max + 1) * 8, &pin); S.off_pinned = pin; S.off_cap = S.off ? nrec_max + 1 : 0; ok = S.off != nullptr;
        if (ok) { const size_t cap = nrec_max * 128 + 64 + (nrec_max * 128) / 2 + (1u << 20); S.seq = (uint8_t*)host_alloc(cap, &pin); S.pinned = pin; S.seq_cap = S.seq ? cap : 0; ok = S.seq != nullptr; }
        { std::lock_guard<std::mutex> lk(amu); alloc_state[i] = ok ? 1 : 2; } acv.notify_all();
      }
    });
  }
  bool wait_slot(size_t i) { if (alloc_state.empty()) return true; std::unique_lock<std::mutex> lk(amu); if (i > alloc_asked) { alloc_asked = i; acv.notify_all(); } acv.wait(lk, [&] { return alloc_state[i] != 0; }); return alloc_state[i] == 1; }
  ~sq_reader() {
    if (prealloc.joinable()) { { std::lock_guard<std::mutex> lk(amu); alloc_stop = true; } acv.notify_all(); prealloc.join(); }
    if (dev) sq_dev_reader_close(dev);
    for (int i = 0; i < 2; ++i) { q[i].finish(); cq[i].finish(); if (th[i].joinable()) th[i].join(); }
    pool.reset();   // after the stream threads: no more tasks are submitted
    for (auto& s : slots) { host_free(s.seq, s.pinned); host_free(s.off, s.off_pinned); }
  }
  // next record of stream i -> (ptr, len); false at end of stream
  bool rec(int i, const char** p, uint32_t* l, const char** nm = nullptr, uint32_t* nl = nullptr) {
    for (;;) {
      if (cur[i] && cur_rec[i] < cur[i]->len.size()) {
        *l = cur[i]->len[cur_rec[i]];
        *p = cur[i]->seq.data() + cur_byte[i];
        cur_byte[i] += *l;
        if (nm && keep_names) { *nl = cur[i]->nlen[cur_rec[i]]; *nm = cur[i]->names.data() + cur_nbyte[i]; cur_nbyte[i] += *nl; }
        ++cur_rec[i];
        return true;
      }
      cur[i] = q[i].get(); cur_rec[i] = 0; cur_byte[i] = 0; cur_nbyte[i] = 0;
      if (!cur[i]) return false;
    }
  }
};

extern "C" int sq_reader_open_ex(const char* const* files1, uint32_t n1, const char* const* files2, uint32_t n2, uint32_t batch_reads,
    uint32_t num_slots, uint32_t flags, sq_reader** out) {
  if (!files1 || n1 == 0 || !out || batch_reads == 0 || (n2 && !files2)) {
    sq_set_error("sq_reader_open: bad arguments");
    return SQ_ERR_ARG;
  }
  if (n2 && n2 != n1) { sq_set_error("sq_reader_open: %u mate-1 files but %u mate-2 files", n1, n2); return SQ_ERR_ARG; }
  std::unique_ptr<sq_reader> R(new sq_reader()); R->paired = n2 > 0; R->batch = batch_reads; R->keep_names = (flags & SQ_READER_KEEP_NAMES) != 0;
  const bool kn = R->keep_names;
  R->slots.resize(num_slots < 2 ? 2 : (num_slots > 8 ? 8 : num_slots));
  std::vector<std::string> a(files1, files1 + n1), b; if (n2) b.assign(files2, files2 + n2);
  // fast path for 4-line FASTQ (the first record of every file is looked at); anything else takes the kseq-rules path
  bool fast = !getenv("SQ_READER_SAFE");
  // Only regular files are probed (and later mapped / reopened): a FIFO or /dev/fd/N from process substitution — the usual way to feed
  // salmon from a decompressor — can be opened once and read once, so those go straight to the streaming path, untouched.
  for (int st = 0; st < 2 && fast; ++st) for (const auto& path : (st ? b : a)) {
    struct stat sb; if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) { fast = false; break; }
  }
  for (int st = 0; st < 2 && fast; ++st) for (const auto& path : (st ? b : a)) {
    std::vector<char> head(65536); gzFile f = gzopen(path.c_str(), "rb"); if (!f) { fast = false; break; }   // the stream thread reports it
    const int n = gzread(f, head.data(), (unsigned)head.size()); gzclose(f);
    if (n > 0 && !looks_like_simple_fastq(head.data(), (size_t)n)) { fast = false; break; }
  }
  R->fast = fast;
  // [r4] device-side record splitting: regular 4-line FASTQ files, no read names wanted, a device present, SQ_READER_DEVICE != 0.  [r5] gzip / BGZF files
  // too (all of them compressed, or none): the device reader inflates them into its ring itself (hip/fastq_dev.hip); SQ_READER_DEVICE_GZ=0 keeps them here
  bool plain = fast && !kn && !(getenv("SQ_READER_DEVICE") && atoi(getenv("SQ_READER_DEVICE")) == 0);
  if (plain && getenv("SQ_READER_DEVICE_GZ") && atoi(getenv("SQ_READER_DEVICE_GZ")) == 0) {   // compressed files stay on the host path: look for the gzip magic
    for (int st = 0; st < 2 && plain; ++st) for (const auto& path : (st ? b : a)) {
      unsigned char mg[2] = {0, 0}; FILE* f = fopen(path.c_str(), "rb"); if (!f) { plain = false; break; }
      const size_t got = fread(mg, 1, 2, f); fclose(f); if (got == 2 && mg[0] == 0x1f && mg[1] == 0x8b) { plain = false; break; }
    }
  }
  if (plain) {
    const int rc = sq_dev_reader_open(a, b, batch_reads, (uint32_t)R->slots.size(), &R->dev);
    if (rc == SQ_OK) { *out = R.release(); return SQ_OK; }
    if (rc != SQ_ERR_DEVICE) return rc;   // no device: the host path below
    R->dev = nullptr;
  }
  R->start_prealloc();
  if (fast) {
    unsigned nt = getenv("SQ_READER_THREADS") ? (unsigned)atoi(getenv("SQ_READER_THREADS")) : std::min(32u, std::max(2u, std::thread::hardware_concurrency() / 2));
    R->pool.reset(new Pool(std::max(1u, nt)));
    R->th[0] = std::thread(produce_fast, a, &R->cq[0], R->pool.get(), kn);
    if (n2) R->th[1] = std::thread(produce_fast, b, &R->cq[1], R->pool.get(), false);
  } else {
    R->th[0] = std::thread(produce, a, &R->q[0], kn);
    if (n2) R->th[1] = std::thread(produce, b, &R->q[1], false);
  }
  *out = R.release();
  return SQ_OK;
}

extern "C" int sq_reader_open(const char* const* files1, uint32_t n1, const char* const* files2, uint32_t n2, uint32_t batch_reads,
    uint32_t num_slots, sq_reader** out) {
  return sq_reader_open_ex(files1, n1, files2, n2, batch_reads, num_slots, 0, out);
}

// Fills *b with the next batch (b->n == 0 at the end of input).  The arrays live in slot *slot of the reader and stay
// valid until sq_reader_release(reader, slot); with S slots the caller may hold S-1 batches while asking for the next.
extern "C" int sq_reader_next(sq_reader* R, sq_read_batch* b, int* slot) {
  if (!R || !b || !slot) { sq_set_error("sq_reader_next: bad arguments"); return SQ_ERR_ARG; }
  memset(b, 0, sizeof(*b)); *slot = -1; b->paired = R->paired ? 1 : 0;
  if (R->dev) return sq_dev_reader_next(R->dev, b, slot);
  if (R->ended) return SQ_OK;
  int si = -1; for (size_t i = 0; i < R->slots.size(); ++i) if (!R->slots[i].busy) { si = (int)i; break; }
  if (si < 0) {
    sq_set_error("sq_reader_next: all %zu batch buffers are in use (sq_reader_release one first)", R->slots.size());
    return SQ_ERR_STATE;
  }
  Slot& S = R->slots[(size_t)si];
  const size_t nrec_max = (size_t)R->batch * (R->paired ? 2 : 1);
  if (!R->wait_slot((size_t)si)) { sq_set_error("sq_reader: out of memory"); return SQ_ERR_NOMEM; }
  if (S.off_cap < nrec_max + 1) {
    host_free(S.off, S.off_pinned);
    bool pin = false;
    S.off = (uint64_t*)host_alloc((nrec_max + 1) * 8, &pin);
    S.off_pinned = pin;
    S.off_cap = nrec_max + 1;
    if (!S.off) {
      sq_set_error("sq_reader: out of memory");
      return SQ_ERR_NOMEM;
    }
  }
  auto grow = [&](size_t need) -> bool {
    if (need <= S.seq_cap) return true;
    size_t cap = need + need / 2 + (1u << 20); bool pin; uint8_t* nb = (uint8_t*)host_alloc(cap, &pin); if (!nb) return false;
    if (S.seq) { memcpy(nb, S.seq, S.seq_cap); host_free(S.seq, S.pinned); }
    S.seq = nb; S.seq_cap = cap; S.pinned = pin; return true;
  };
  if (!grow((size_t)nrec_max * 128 + 64)) { sq_set_error("sq_reader: out of memory"); return SQ_ERR_NOMEM; }
  if (R->fast) {
    std::vector<sq_reader::Seg> sg[2];
    const int ns = R->paired ? 2 : 1;
    size_t got[2] = {0, 0};
    for (int i = 0; i < ns; ++i) got[i] = R->gather(i, R->batch, sg[i]);
    for (int i = 0; i < ns; ++i) { std::string e; { std::lock_guard<std::mutex> lk(R->cq[i].mu); e = R->cq[i].err; } if (!e.empty()) { R->ended = true; sq_set_error("%s", e.c_str()); return SQ_ERR_IO; } }
    size_t n = got[0];
    if (R->paired && got[0] != got[1]) {   // one stream ended early
      R->ended = true;
      sq_set_error("mate files have different numbers of records (stopped after %llu pairs)", (unsigned long long)(R->total + std::min(got[0], got[1])));
      return SQ_ERR_IO;
    }
    if (n < R->batch) R->ended = true;
    if (n == 0) return SQ_OK;
    // flat per-stream views of the batch's records: segment prefix counts
    std::vector<size_t> pre[2]; for (int i = 0; i < ns; ++i) { pre[i].assign(sg[i].size() + 1, 0); for (size_t k = 0; k < sg[i].size(); ++k) pre[i][k + 1] = pre[i][k] + sg[i][k].count; }
    const unsigned K = (unsigned)std::max<size_t>(1, std::min<size_t>(R->pool->th.size(), n / 4096 + 1));
    std::vector<uint64_t> part_bytes(K + 1, 0);
    auto walk = [&](unsigned j, bool fill) {
      const size_t a = n * j / K, b2 = n * (j + 1) / K;
      size_t k[2] = {0, 0}; for (int i = 0; i < ns; ++i) while (pre[i][k[i] + 1] <= a && k[i] + 1 < sg[i].size()) ++k[i];
      uint64_t bytes = fill ? part_bytes[j] : 0;
      for (size_t r = a; r < b2; ++r) {
        for (int i = 0; i < ns; ++i) {
          while (r >= pre[i][k[i] + 1]) ++k[i];
          const sq_reader::Seg& g = sg[i][k[i]]; const size_t x = g.first + (r - pre[i][k[i]]);
          const uint32_t l = g.c->len[x];
          if (fill) { memcpy(S.seq + bytes, g.c->text + g.c->pos[x], l); S.off[(R->paired ? 2 * r : r) + (size_t)i + 1] = bytes + l; }
          bytes += l;
        }
      }
      if (!fill) part_bytes[j + 1] = bytes;
    };
    R->pool->parallel(K, [&](unsigned j) { walk(j, false); });
    { uint64_t acc = 0; for (unsigned j = 0; j < K; ++j) { const uint64_t v = part_bytes[j + 1]; part_bytes[j] = acc; acc += v; } part_bytes[K] = acc; }
    const uint64_t bytes = part_bytes[K];
    if (!grow(bytes + 64)) { sq_set_error("sq_reader: out of memory"); return SQ_ERR_NOMEM; }
    S.off[0] = 0;
    R->pool->parallel(K, [&](unsigned j) { walk(j, true); });
    memset(S.seq + bytes, 0, 16);
    if (R->keep_names) {   // names of the mate-1 stream, in batch order (serial: only the SAM writer asks for them)
      S.names.clear(); S.name_off.assign(1, 0);
      for (const auto& g : sg[0]) for (size_t x = g.first; x < g.first + g.count; ++x) {
        S.names.insert(S.names.end(), g.c->text + g.c->npos[x], g.c->text + g.c->npos[x] + g.c->nlen[x]); S.name_off.push_back(S.names.size());
      }
    }
    for (int i = 0; i < ns; ++i) R->consume(i, n);
    S.busy = true; R->total += n;
    b->n = (uint32_t)n; b->seq = S.seq; b->seq_off = S.off; b->on_device = 0; *slot = si;
    return SQ_OK;
  }
  uint32_t n = 0; uint64_t bytes = 0; S.off[0] = 0;
  if (R->keep_names) { S.names.clear(); S.name_off.assign(1, 0); }
  while (n < R->batch) {
    const char* p1; uint32_t l1; const char* p2 = nullptr; uint32_t l2 = 0; const char* nm = nullptr; uint32_t nl = 0;
    const bool h1 = R->rec(0, &p1, &l1, &nm, &nl); const bool h2 = R->paired ? R->rec(1, &p2, &l2) : h1;
    if (!h1 || !h2) {
      R->ended = true;
      for (int i = 0; i < (R->paired ? 2 : 1); ++i) {
        std::lock_guard<std::mutex> lk(R->q[i].mu);
        if (!R->q[i].err.empty()) {
          sq_set_error("%s", R->q[i].err.c_str());
          return SQ_ERR_IO;
        }
      }
      if (R->paired && h1 != h2) {
        sq_set_error("mate files have different numbers of records (stopped after %llu pairs)", (unsigned long long)(R->total + n));
        return SQ_ERR_IO;
      }
      break;
    }
    if (!grow(bytes + l1 + l2 + 64)) { sq_set_error("sq_reader: out of memory"); return SQ_ERR_NOMEM; }
    memcpy(S.seq + bytes, p1, l1); bytes += l1; S.off[R->paired ? 2 * n + 1 : n + 1] = bytes;
    if (R->keep_names) { S.names.insert(S.names.end(), nm, nm + nl); S.name_off.push_back(S.names.size()); }
    if (R->paired) { memcpy(S.seq + bytes, p2, l2); bytes += l2; S.off[2 * n + 2] = bytes; }
    ++n;
  }
  if (n == 0) return SQ_OK;
  memset(S.seq + bytes, 0, 16);    // the packing kernel reads whole 4-byte words
  S.busy = true; R->total += n;
  b->n = n; b->seq = S.seq; b->seq_off = S.off; b->on_device = 0; *slot = si;
  return SQ_OK;
}
extern "C" void sq_reader_release(sq_reader* R, int slot) {
  if (R && R->dev) { sq_dev_reader_release(R->dev, slot); return; }
  if (R && slot >= 0 && (size_t)slot < R->slots.size()) R->slots[(size_t)slot].busy = false;
}
// names of the batch in `slot` (reader opened with SQ_READER_KEEP_NAMES): name i = names[name_off[i] .. name_off[i+1])
extern "C" int sq_reader_names(const sq_reader* R, int slot, const char** names, const uint64_t** name_off) {
  if (!R || !names || !name_off || slot < 0 || (size_t)slot >= R->slots.size()) { sq_set_error("sq_reader_names: bad arguments"); return SQ_ERR_ARG; }
  if (!R->keep_names) { sq_set_error("sq_reader_names: the reader was opened without SQ_READER_KEEP_NAMES"); return SQ_ERR_STATE; }
  const Slot& S = R->slots[(size_t)slot]; *names = S.names.data(); *name_off = S.name_off.data(); return SQ_OK;
}
extern "C" uint64_t sq_reader_total(const sq_reader* R) { return R ? (R->dev ? sq_dev_reader_total(R->dev) : R->total) : 0; }
extern "C" void sq_reader_close(sq_reader* R) { delete R; }
