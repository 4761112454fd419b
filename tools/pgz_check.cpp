// tools/pgz_check.cpp — TEST TOOL.  Inflates a gzip file twice, through zlib (one stream) and through host/pgzip.cpp (pieces on a small
// thread pool), and compares the bytes:  pgz_check file.gz [threads] [piece_bytes]  ->  exit 0 and "equal=1 ..." when they agree.
// tests/test_pgzip.py drives it over compression levels, flush points, several members, stored blocks, binary data and damaged files.
#include "../salmon_amd/csrc/host/pgzip.h"
#include <zlib.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
struct Pool {
  std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; std::vector<std::thread> th; bool stop = false;
  explicit Pool(unsigned n) { for (unsigned i = 0; i < n; ++i) th.emplace_back([this] { for (;;) { std::function<void()> f;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); } f(); } }); }
  ~Pool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
  void submit(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_one(); }
};
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: pgz_check file.gz [threads] [piece_bytes]\n"); return 2; }
  const char* path = argv[1]; const unsigned T = argc > 2 ? (unsigned)atoi(argv[2]) : 4; const size_t piece = argc > 3 ? (size_t)atol(argv[3]) : (1u << 20);
  int fd = open(path, O_RDONLY); struct stat sb; if (fd < 0 || fstat(fd, &sb) != 0) { perror(path); return 2; }
  const uint8_t* m = (const uint8_t*)mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  std::vector<char> ref; bool zerr = false;
  { gzFile f = gzopen(path, "rb"); gzbuffer(f, 1 << 20); std::vector<char> b(1 << 22); int n; while ((n = gzread(f, b.data(), (unsigned)b.size())) > 0) ref.insert(ref.end(), b.data(), b.data() + n);
    int e = 0; gzerror(f, &e); zerr = n < 0 || (e != Z_OK && e != Z_STREAM_END); gzclose(f); }
  Pool pool(T);
  PgzStream* s = pgz_open(m, (size_t)sb.st_size, [&](std::function<void()> f) { pool.submit(std::move(f)); }, T, piece);
  if (!s) { printf("not a gzip file\n"); return 3; }
  std::vector<char> out; std::string err; long n;
  if (argc > 4) { PgzBuf B; int rc; while ((rc = pgz_next(s, &B, &err)) > 0) out.insert(out.end(), B.p, B.p + B.n); n = rc; }   // the zero-copy interface
  else { std::vector<char> b(1 << 22); while ((n = pgz_read(s, b.data(), b.size(), &err)) > 0) out.insert(out.end(), b.data(), b.data() + n); }
  const pgz_counters c = pgz_stats(s); pgz_close(s);
  if (n < 0) { printf("pgz error: %s (zlib %s)\n", err.c_str(), zerr ? "failed too" : "read it"); return zerr ? 4 : 1; }
  const bool same = out.size() == ref.size() && memcmp(out.data(), ref.data(), out.size()) == 0;
  printf("equal=%d bytes=%zu pieces=%llu resynced=%llu members=%llu rounds=%llu zlib_error=%d\n", (int)same, out.size(), (unsigned long long)c.pieces, (unsigned long long)c.resynced,
         (unsigned long long)c.members, (unsigned long long)c.rounds, (int)zerr);
  return same ? 0 : 1;
}
