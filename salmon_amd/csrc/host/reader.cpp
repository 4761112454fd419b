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
#include "index.h"
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct RecBlock { std::vector<char> seq; std::vector<uint32_t> len; };   // sequences back to back

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
void produce(std::vector<std::string> files, BlockQueue* out) {
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

struct Slot { uint8_t* seq = nullptr; size_t seq_cap = 0; uint64_t* off = nullptr; size_t off_cap = 0; bool pinned = false,
    off_pinned = false; bool busy = false; };

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
  bool paired = false; uint32_t batch = 0; BlockQueue q[2]; std::thread th[2];
  std::unique_ptr<RecBlock> cur[2]; size_t cur_rec[2] = {0, 0}, cur_byte[2] = {0, 0};
  std::vector<Slot> slots; uint64_t total = 0; bool ended = false;
  ~sq_reader() {
    for (int i = 0; i < 2; ++i) { q[i].finish(); if (th[i].joinable()) th[i].join(); }
    for (auto& s : slots) { host_free(s.seq, s.pinned); host_free(s.off, s.off_pinned); }
  }
  // next record of stream i -> (ptr, len); false at end of stream
  bool rec(int i, const char** p, uint32_t* l) {
    for (;;) {
      if (cur[i] && cur_rec[i] < cur[i]->len.size()) {
        *l = cur[i]->len[cur_rec[i]];
        *p = cur[i]->seq.data() + cur_byte[i];
        cur_byte[i] += *l;
        ++cur_rec[i];
        return true;
      }
      cur[i] = q[i].get(); cur_rec[i] = 0; cur_byte[i] = 0;
      if (!cur[i]) return false;
    }
  }
};

extern "C" int sq_reader_open(const char* const* files1, uint32_t n1, const char* const* files2, uint32_t n2, uint32_t batch_reads,
    uint32_t num_slots, sq_reader** out) {
  if (!files1 || n1 == 0 || !out || batch_reads == 0 || (n2 && !files2)) {
    sq_set_error("sq_reader_open: bad arguments");
    return SQ_ERR_ARG;
  }
  if (n2 && n2 != n1) { sq_set_error("sq_reader_open: %u mate-1 files but %u mate-2 files", n1, n2); return SQ_ERR_ARG; }
  std::unique_ptr<sq_reader> R(new sq_reader()); R->paired = n2 > 0; R->batch = batch_reads;
  R->slots.resize(num_slots < 2 ? 2 : (num_slots > 8 ? 8 : num_slots));
  std::vector<std::string> a(files1, files1 + n1), b; if (n2) b.assign(files2, files2 + n2);
  R->th[0] = std::thread(produce, a, &R->q[0]);
  if (n2) R->th[1] = std::thread(produce, b, &R->q[1]);
  *out = R.release();
  return SQ_OK;
}

// Fills *b with the next batch (b->n == 0 at the end of input).  The arrays live in slot *slot of the reader and stay
// valid until sq_reader_release(reader, slot); with S slots the caller may hold S-1 batches while asking for the next.
extern "C" int sq_reader_next(sq_reader* R, sq_read_batch* b, int* slot) {
  if (!R || !b || !slot) { sq_set_error("sq_reader_next: bad arguments"); return SQ_ERR_ARG; }
  memset(b, 0, sizeof(*b)); *slot = -1; b->paired = R->paired ? 1 : 0;
  if (R->ended) return SQ_OK;
  int si = -1; for (size_t i = 0; i < R->slots.size(); ++i) if (!R->slots[i].busy) { si = (int)i; break; }
  if (si < 0) {
    sq_set_error("sq_reader_next: all %zu batch buffers are in use (sq_reader_release one first)", R->slots.size());
    return SQ_ERR_STATE;
  }
  Slot& S = R->slots[(size_t)si];
  const size_t nrec_max = (size_t)R->batch * (R->paired ? 2 : 1);
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
  if (!grow((size_t)nrec_max * 160 + 64)) { sq_set_error("sq_reader: out of memory"); return SQ_ERR_NOMEM; }
  uint32_t n = 0; uint64_t bytes = 0; S.off[0] = 0;
  while (n < R->batch) {
    const char* p1; uint32_t l1; const char* p2 = nullptr; uint32_t l2 = 0;
    const bool h1 = R->rec(0, &p1, &l1); const bool h2 = R->paired ? R->rec(1, &p2, &l2) : h1;
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
  if (R && slot >= 0 && (size_t)slot < R->slots.size()) R->slots[(size_t)slot].busy = false;
}
extern "C" uint64_t sq_reader_total(const sq_reader* R) { return R ? R->total : 0; }
extern "C" void sq_reader_close(sq_reader* R) { delete R; }
