// host/bgzf_source.h — the worker pool of the host read pipeline and the parallel BGZF source, shared by the FASTQ reader (reader.cpp) and the
// alignment reader (sam_reader.cpp: BAM files, bgzipped SAM).
#pragma once
#include <zlib.h>
#include <sys/mman.h>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "crc32_fast.h"
#include "pgzip.h"

namespace sqio {
struct Pool {   // a few workers shared by the parse tasks of both streams and the batch assembly
  std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; std::vector<std::thread> th; bool stop = false;
  explicit Pool(unsigned n) { for (unsigned i = 0; i < n; ++i) th.emplace_back([this] { run(); }); }
  ~Pool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
  void run() { for (;;) { std::function<void()> f; { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); } f(); } }
  void submit(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_one(); }
  // run fn(0..n-1) on the pool and wait
  void parallel(unsigned n, const std::function<void(unsigned)>& fn) {
    if (n <= 1) { if (n) fn(0); return; }
    std::mutex m; std::condition_variable c; unsigned left = n;
    for (unsigned i = 0; i < n; ++i) submit([&, i] { fn(i); std::lock_guard<std::mutex> lk(m); if (--left == 0) c.notify_one(); });
    std::unique_lock<std::mutex> lk(m); c.wait(lk, [&] { return left == 0; });
  }
};

struct Mapping { void* p = nullptr; size_t n = 0; ~Mapping() { if (p) munmap(p, n); } };

// ---- BGZF (bgzip / htslib): a gzip file made of members of at most 64 KB that each name their compressed size in a 'BC' extra field.
// A plain gzip stream can only be inflated by one thread; these members are independent, so the pool inflates them in parallel and the
// stream thread takes the text in file order.  (The reference reads .gz through one zlib stream per file: include/salmon/internal/io/FastxReader.hpp.)
struct BgzfSource {
  // [r4] The members are inflated in GROUPS (about 4 MB of text: every member says how much it holds in its trailer) straight into the group's buffer,
  // each at its own offset, by one pool task per group; the stream thread takes the groups in file order and parses their text where it lies.  (Before,
  // every 64 KB member was a task and the stream thread copied the members' text into parse chunks one by one: that one thread per file was the bound,
  // 2.6 GB/s.)
  struct Mem { size_t off, csize; uint32_t isize; size_t at; };
  struct Grp { std::vector<Mem> mem; size_t total = 0; std::unique_ptr<char[]> text; bool done = false; std::string err; };
  std::shared_ptr<Mapping> map; const uint8_t* base = nullptr; size_t n = 0, next_off = 0; Pool* pool = nullptr;
  std::mutex mu; std::condition_variable cv; std::deque<std::shared_ptr<Grp>> win; size_t window = 16; std::string path; std::string err;
  static constexpr size_t GROUP_TEXT = 4u << 20; static constexpr size_t GROUP_MEMBERS = 512;
  // size of the member starting at p (0: not a BGZF member)
  static size_t member_size(const uint8_t* p, size_t left) {
    if (left < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8); if (12 + xlen > left) return 0;
    for (size_t q = 12; q + 4 <= 12 + xlen;) {
      const size_t slen = (size_t)p[q + 2] | ((size_t)p[q + 3] << 8);
      if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) return ((size_t)p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
      q += 4 + slen;
    }
    return 0;
  }
  static uint32_t le32(const uint8_t* t) { return (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24); }
  // one member into dst (its isize bytes); "" or what is wrong with it
  static const char* inflate_member(const uint8_t* p, const Mem& m, char* dst) {
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8); const size_t hdr = 12 + xlen;
    if (m.csize < hdr + 8 || (m.isize && m.csize == hdr + 8)) return "truncated BGZF member";   // (text without a deflate stream in front of the trailer)
    const uint32_t crc = le32(p + m.csize - 8);
    if (!m.isize) return "";
    if (pgz_inflate_raw(p + hdr, m.csize - hdr - 8, dst, m.isize) != (long)m.isize) return "corrupt BGZF member";      // the own inflate in byte mode (host/pgzip.cpp; measured against zlib's in round 4: 510 against 283 MB/s per thread)
    if (sqcrc::crc32((uint32_t)crc32(0L, Z_NULL, 0), dst, m.isize) != crc) return "BGZF checksum mismatch";
    return "";
  }
  void schedule() {   // caller holds mu
    while (win.size() < window && next_off < n) {
      auto g = std::make_shared<Grp>();
      while (next_off < n && g->total < GROUP_TEXT && g->mem.size() < GROUP_MEMBERS) {
        const size_t ms = member_size(base + next_off, n - next_off);
        if (ms == 0 || ms > n - next_off || ms < 26) { if (g->mem.empty()) { g->err = ms > n - next_off ? "truncated BGZF member" : (ms ? "truncated BGZF member" : "not a BGZF member (mixed gzip file?)"); g->done = true; } next_off = g->mem.empty() ? n : next_off; break; }
        const uint32_t isize = le32(base + next_off + ms - 4);
        if (isize > (1u << 16)) { if (g->mem.empty()) { g->err = "BGZF member larger than 64 KB"; g->done = true; next_off = n; } break; }
        g->mem.push_back(Mem{next_off, ms, isize, g->total}); g->total += isize; next_off += ms;
      }
      if (g->mem.empty() && !g->done) break;   // (a damaged member right behind a full group: the next call reports it)
      win.push_back(g);
      if (g->done) break;
      // notify while holding mu: once `done` is visible the destructor may run, and it must not free cv under a task still about to signal it
      pool->submit([this, g] {
        std::string e;
        g->text.reset(new (std::nothrow) char[g->total + 64]);   // + slack: the own inflate copies matches 32 bytes at a time
        if (!g->text) e = "out of memory";
        else for (const Mem& m : g->mem) { const char* w = inflate_member(base + m.off, m, g->text.get() + m.at); if (*w) { e = w; break; } }
        std::lock_guard<std::mutex> lk(mu); g->err = e; g->done = true; cv.notify_all(); });
    }
  }
  // the next group's text, in file order: 1 and *out; 0 at the end; -1 on error (see err)
  int next_buf(PgzBuf* out) {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      schedule();
      if (win.empty()) return 0;
      auto g = win.front();
      cv.wait(lk, [&] { return g->done; });
      win.pop_front();
      if (!g->err.empty()) { err = "'" + path + "': " + g->err; return -1; }
      if (!g->total) continue;   // empty members only (the end-of-file marker)
      out->p = g->text.get(); out->n = g->total; out->hold = std::shared_ptr<void>(g, (void*)g.get());
      return 1;
    }
  }
  ~BgzfSource() {   // let the queued tasks finish: they hold `this`
    std::unique_lock<std::mutex> lk(mu);
    for (auto& g : win) cv.wait(lk, [&] { return g->done; });
  }
};

}  // namespace sqio
