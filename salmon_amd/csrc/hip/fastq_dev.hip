// hip/fastq_dev.hip — [r4] FASTQ record splitting on the device (the north_star's "newline ballot + segmented scan").
//
// The host read pipeline (host/reader.cpp: mmap + memchr per line + a gather into page-locked batches) tops out near 33 M pairs/s whatever the
// number of worker threads (page faults of one address space, two passes over every record); the GPU path behind it maps > 200 M pairs/s.  Here
// the host only moves TEXT: the bytes of the next batch's records of each mate file are pread() into a page-locked buffer by a few threads (which
// count newlines as they go, to cut exactly 4 n lines), copied to HBM, and the device finds the records itself:
//   k_fq_count    newlines per 4 KB tile (16 bytes per lane, zero-byte SWAR on x ^ 0x0A0A0A0A)
//   scan          tile bases (scan_kernels.h)
//   k_fq_index    position of every newline (block-exclusive scan of the lanes' counts + the tile base)
//   k_fq_records  record r = lines 4r .. 4r+3: '@' / sequence / '+' / quality of the same length (CR stripped), length + start of the sequence
//   scan          read offsets of the interleaved batch (mate 1, mate 2, mate 1, ...)
//   k_fq_copy     the bases into the compact buffer sq_map_batch takes (on_device = 1)
// Two producer threads keep the slots filled: one stages text (parallel pread into a ring of page-locked pieces + newline counts), one uploads the
// pieces and splits (each slot on its own stream), so reading, H2D, splitting and the mapping of earlier batches overlap.  Plain, regular, 4-line
// FASTQ files only; anything else (gzip, FASTA, wrapped records, FIFOs, read names wanted) stays on the host path.  Replaces, for such input, the
// reference's FastxParser producer threads (include/salmon/internal/io/FastxReader.hpp:13-32, SalmonQuantify.cpp:2419-2443).
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "scan_kernels.h"
#include "../host/index.h"
#include "../host/reader_dev.h"
#include "../host/bgzf_source.h"
#include "inflate_dev.h"
#include "gzip_dev.h"

namespace {
constexpr int FQ_TB = 256;   // 16 bytes per lane: a 4 KB tile per block
__device__ inline uint32_t nl_mask4(uint32_t w) {   // 0x80 in every byte of w that is '\n' (exact zero-byte detection on w ^ 0x0A0A0A0A)
  const uint32_t x = w ^ 0x0A0A0A0Au;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
__global__ void __launch_bounds__(FQ_TB) k_fq_count(const uint4* __restrict__ text, uint64_t nvec, uint32_t* __restrict__ tile_cnt) {
  const uint64_t i = (uint64_t)blockIdx.x * FQ_TB + threadIdx.x;
  uint32_t c = 0;
  if (i < nvec) { const uint4 v = text[i]; c = __popc(nl_mask4(v.x)) + __popc(nl_mask4(v.y)) + __popc(nl_mask4(v.z)) + __popc(nl_mask4(v.w)); }
  for (int s = 32; s >= 1; s >>= 1) c += __shfl_down(c, s, 64);
  __shared__ uint32_t sw[FQ_TB / 64];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < FQ_TB / 64; ++w) t += sw[w]; tile_cnt[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(FQ_TB) k_fq_index(const uint4* __restrict__ text, uint64_t nvec, const uint64_t* __restrict__ tile_base, uint32_t* __restrict__ nlpos, uint64_t cap) {
  __shared__ uint64_t wsum[FQ_TB / 64 + 1];
  const uint64_t i = (uint64_t)blockIdx.x * FQ_TB + threadIdx.x;
  uint32_t m[4] = {0, 0, 0, 0};
  if (i < nvec) { const uint4 v = text[i]; m[0] = nl_mask4(v.x); m[1] = nl_mask4(v.y); m[2] = nl_mask4(v.z); m[3] = nl_mask4(v.w); }
  const uint32_t c = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
  uint64_t tot; uint64_t at = tile_base[blockIdx.x] + sqk::scan_block_excl<FQ_TB>(c, &tot, wsum);
  const uint32_t byte0 = (uint32_t)(i * 16);
#pragma unroll
  for (int w = 0; w < 4; ++w) { uint32_t x = m[w]; while (x) { const int b = (__ffs((int)x) - 1) >> 3; x &= x - 1; if (at < cap) nlpos[at] = byte0 + 4u * w + (uint32_t)b; ++at; } }
}
// err[0] = first bad record (global index within the batch) + 1, err[1] = what was wrong with it
__global__ void k_fq_records(const uint8_t* __restrict__ text, const uint32_t* __restrict__ nlpos, uint32_t n, uint32_t mate, uint32_t stride,
                             uint32_t* __restrict__ len, uint32_t* __restrict__ start, unsigned* __restrict__ err) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= n) return;
  const uint32_t s0 = r ? nlpos[4 * r - 1] + 1 : 0, e0 = nlpos[4 * r], s1 = e0 + 1; uint32_t e1 = nlpos[4 * r + 1]; const uint32_t s2 = e1 + 1, e2 = nlpos[4 * r + 2], s3 = e2 + 1; uint32_t e3 = nlpos[4 * r + 3];
  if (e1 > s1 && text[e1 - 1] == '\r') --e1;
  if (e3 > s3 && text[e3 - 1] == '\r') --e3;
  unsigned what = 0;
  if (text[s0] != '@') what = 1; else if (text[s2] != '+') what = 2; else if (e1 - s1 != e3 - s3) what = 3;
  if (what) { const unsigned old = atomicMin(&err[0], r + 1); if (r + 1 <= old) err[1 + mate] = what; }
  len[(size_t)r * stride + mate] = e1 - s1; start[r] = s1;
}
__global__ void k_fq_copy(const uint8_t* __restrict__ text, const uint32_t* __restrict__ start, const uint64_t* __restrict__ off, uint32_t n, uint32_t mate, uint32_t stride,
                          uint8_t* __restrict__ seq, const unsigned* __restrict__ err) {
  // a group of 8 lanes per record: 8 consecutive bytes per trip (a record's bases are contiguous in the text and in the batch)
  if (err[0] != 0xFFFFFFFFu) return;   // a damaged record: its "lengths" mean nothing (and the batch buffer is sized for sound records); the host reports it
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, l = threadIdx.x & 7; if (g >= n) return;
  const uint64_t o0 = off[(size_t)g * stride + mate], o1 = off[(size_t)g * stride + mate + 1]; const uint32_t L = (uint32_t)(o1 - o0); const uint8_t* s = text + start[g]; uint8_t* d = seq + o0;
  for (uint32_t j = l; j < L; j += 8) d[j] = s[j];
}

__global__ void k_fq_pad(const uint64_t* __restrict__ off, uint32_t nrec, uint8_t* __restrict__ seq, const unsigned* __restrict__ err) {   // 16 zero bytes behind the last base
  if (err[0] == 0xFFFFFFFFu && threadIdx.x < 16) seq[off[nrec] + threadIdx.x] = 0;
}

struct Workers {   // a few threads for pread + newline counting
  std::mutex mu; std::condition_variable cv, cvd; std::deque<std::function<void()>> q; std::vector<std::thread> th; bool stop = false; int busy = 0;
  explicit Workers(unsigned n) { for (unsigned i = 0; i < n; ++i) th.emplace_back([this] { for (;;) { std::function<void()> f; { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); ++busy; } f(); { std::lock_guard<std::mutex> lk(mu); --busy; } cvd.notify_all(); } }); }
  ~Workers() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
  void run(unsigned n, const std::function<void(unsigned)>& fn) { { std::lock_guard<std::mutex> lk(mu); for (unsigned i = 0; i < n; ++i) q.push_back([&fn, i] { fn(i); }); } cv.notify_all(); std::unique_lock<std::mutex> lk(mu); cvd.wait(lk, [&] { return q.empty() && busy == 0; }); }
};
inline uint64_t count_nl(const char* p, size_t n) {
  uint64_t c = 0; size_t i = 0; const __m256i nl = _mm256_set1_epi8('\n');
  for (; i + 32 <= n; i += 32) c += (uint64_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i)), nl)));
  for (; i < n; ++i) c += p[i] == '\n';
  return c;
}
}  // namespace

struct sq_dev_reader {
  int device = 0; uint32_t batch = 0; bool paired = false;
  // a mate stream = its files end to end; a file that does not end with a newline gets one (pad = 1), so records never straddle files
  struct File { std::string path; int fd = -1; uint64_t size = 0, vbase = 0; uint32_t pad = 0; };
  // [r5] a gzip / BGZF file: its text arrives as buffers in file order from the pool that inflates it (BGZF: groups of members, each inflated where it will
  // lie; any other gzip stream: pieces, host/pgzip.cpp) and is copied into the ring like the bytes of a plain file — the device then splits the records.
  // A stream with such a file is SEQUENTIAL: its size is known only at its end (vsize stays at its maximum until then), the stager asks for the text up
  // to an offset (seq_ensure) before the round's pieces are filled from the buffers (vread_seq), and buffers behind the batch cut are let go (seq_trim).
  struct SeqFile {
    std::string path; std::shared_ptr<sqio::Mapping> map; std::unique_ptr<sqio::BgzfSource> bg; PgzStream* pz = nullptr; bool is_bgzf = false;
    ~SeqFile() { if (pz) pgz_close(pz); bg.reset(); }
    int next(PgzBuf* B, std::string* e) {
      if (bg) { const int rc = bg->next_buf(B); if (rc < 0) *e = bg->err; return rc; }
      std::string w; const int rc = pgz_next(pz, B, &w); if (rc < 0) *e = "'" + path + "': " + w; return rc;
    }
  };
  struct SeqChunk { uint64_t voff; PgzBuf b; };
  struct Stream { std::vector<File> files; uint64_t vsize = 0, vpos = 0; double est = 260.0; std::string name;   // name: the last file, for messages
    // BGZF files only (bgz): no buffers in between — the members of a round are inflated straight into its ring pieces.  mem: every non-empty member
    // scanned so far, with the offset of its text in the stream (file = ~0u: the newline a file without a last one gets)
    bool gzdev = false;   // [r6] every file an ordinary gzip file, inflated on the device
    bool bgz = false; struct BzMember { uint32_t file; uint32_t csize; uint64_t coff; uint32_t isize; uint64_t voff; uint32_t crc, hdr; };   // crc: of the text (trailer); hdr: bytes of gzip header in front of the deflate stream
    std::vector<BzMember> mem; size_t scan_file = 0; uint64_t scan_off = 0, scan_voff = 0, scan_file_bytes = 0; bool scan_done = false;
    bool seq = false; std::vector<std::unique_ptr<SeqFile>> sfiles; size_t sfile_cur = 0; std::deque<SeqChunk> win; uint64_t wend = 0, file_bytes = 0; bool final_ = false; char last_byte = '\n'; } sm[2];
  std::unique_ptr<sqio::Pool> zpool;   // the inflating threads of the buffered sequential streams
  bool any_gzdev = false;   // [r6] the streams are ordinary gzip files inflated on the device (all of them, or none)
  bool any_buffered = false, any_bgz = false, dev_inflate = false;   // dev_inflate: every stream is BGZF and the members are inflated by hip/inflate_dev.hip
  // [r5] BGZF inflated on the device.  The stager thread feeds each mate stream's CHUNKS: the next members of the stream (256 MB of text or more: ~4000 members of the usual
  // size, a wave each — a launch that fills the chip), their compressed bytes copied from the file mapping into ring pieces, sent to the device, inflated there into the
  // chunk's buffer (a ring of DV_CHUNKS buffers per stream).  The splitter thread cuts batches out of that text: it copies what it expects a batch to take from the
  // chunk buffers into the slot's text (device to device), counts lines there, takes more if the records were longer than expected, and moves the stream's
  // position behind the batch's last line; a chunk whose text is all behind the position goes back to the stager.
  static constexpr int DV_CHUNKS = 6; uint32_t DV_MEMBERS = 32768; uint64_t DV_TEXT = 256u << 20;   // a chunk ends at DV_TEXT bytes of text or DV_MEMBERS members, whichever comes first
  struct DvChunk { int idx = 0; uint64_t voff = 0; size_t n = 0; bool last = false; std::vector<std::pair<uint32_t, uint64_t>> where; };   // where: (file, offset) of its members, for messages
  struct DvStream {
    void* text[DV_CHUNKS] = {}; size_t text_cap[DV_CHUNKS] = {}; void* comp[DV_CHUNKS] = {}; size_t comp_cap[DV_CHUNKS] = {}; void* mem[DV_CHUNKS] = {}; size_t mem_cap[DV_CHUNKS] = {};
    uint32_t* st = nullptr;                      // 2 words per chunk buffer (inflate_dev.hip's status)
    hipStream_t hs[2] = {nullptr, nullptr}; hipEvent_t ev_h2d[DV_CHUNKS] = {}, ev_done[DV_CHUNKS] = {};
    std::deque<DvChunk> q;                        // inflated or being inflated, in stream order (mu)
    std::deque<std::pair<int, std::vector<int>>> lent;   // (chunk buffer, ring pieces) whose copies to the device may still run (stager only)
    uint64_t produced = 0, ahead = 0; size_t next_mem = 0; bool finished = false;   // stager; `ahead`: the stream offset behind the last chunk made
    // [r6] ordinary gzip files inflated on the device (hip/gzip_dev.hip): the decoder of the file being read, which file that is, its text so far and its last byte
    sq_gzdev* gz = nullptr; size_t gz_file = 0; uint64_t gz_file_bytes = 0; char gz_last = '\n'; const char* gz_last_at = nullptr;   // gz_last_at: where the file's last byte so far lies (device)
    uint64_t pos = 0;                             // splitter: the stream offset of the next batch's first byte
  } dv[2];
  double t_dv_fill = 0, t_dv_wait = 0, t_dv_split = 0;
  struct Slot {   // device side only: the text of a batch never sits in host memory as a whole
    void* d_text[2] = {nullptr, nullptr}; size_t text_cap[2] = {0, 0};
    void* d_nlpos[2] = {nullptr, nullptr}; size_t nl_cap[2] = {0, 0};
    void* d_start[2] = {nullptr, nullptr}; size_t start_cap[2] = {0, 0};
    void* d_tile = nullptr; size_t tile_cap = 0; void* d_tbase = nullptr; size_t tbase_cap = 0; void* d_spine = nullptr; size_t spine_cap = 0;
    void* d_len = nullptr; size_t len_cap = 0; void* d_off = nullptr; size_t off_cap = 0;
    void* d_seq = nullptr; size_t seq_cap = 0; unsigned* d_err = nullptr; unsigned* h_res = nullptr; hipStream_t st = nullptr;
    uint32_t n = 0; size_t bytes[2] = {0, 0};
  };
  std::vector<Slot> slots;
  std::unique_ptr<Workers> pool;
  // [r4] The page-locked memory is a RING of pieces (PIECE bytes each, RING_PIECES of them: 256 MB whatever the batch size — page-locking memory
  // costs ~0.2 s per GB, as much as reading it).  The stager fills a ROUND of up to ROUND_PIECES pieces at a time (parallel pread + newline
  // counts), finds where the batch ends, and hands the round to the uploader, which copies the pieces behind each other into the slot's text
  // buffer and gives them back; once a mate's text is complete its splitting kernels go out.
  static constexpr size_t PIECE = 4u << 20; int RING_PIECES = 64, ROUND_PIECES = 16;   // [r5] compressed input: 96 / 32 (a round is then 128 MB of text: ~500 tasks for the inflating pool)
  char* ring = nullptr; std::deque<int> free_pieces;
  struct Round {
    int slot = -1, mate = 0; std::vector<std::pair<int, size_t>> pieces;   // (ring piece, valid bytes)
    size_t dst = 0;                       // where this round's first byte goes in the mate's text
    bool first_of_batch = false, last_of_mate = false, last_of_batch = false, end_of_input = false; uint32_t n = 0; size_t total = 0;   // n, total: valid with last_of_mate
    int rc = SQ_OK; std::string err;
  };
  // stager -> (rounds) -> uploader -> (ready) -> consumer -> (free_slots) -> stager
  std::thread prod, prod2; std::mutex mu; std::condition_variable cv; std::deque<int> ready, free_slots; std::deque<Round> rounds; bool stop = false, done = false;
  std::string err; int err_rc = SQ_OK; uint64_t total = 0, staged_total = 0;
  double t_stage = 0, t_upload = 0, t_wait_piece = 0, t_wait_text = 0, t_fill = 0; uint64_t text_bytes = 0;   // SQ_READER_STATS=1 prints them at close

  static int dev_grow(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0; const size_t c = need + need / 4 + 4096;
    if (hipMalloc(p, c) != hipSuccess) { (void)hipGetLastError(); return -1; }
    *cap = c; return 0;
  }
  static int dev_grow_keep(void** p, size_t* cap, size_t need, size_t keep, hipStream_t st) {   // as dev_grow, the first `keep` bytes survive
    if (need <= *cap) return 0;
    void* nb = nullptr; const size_t c = need + need / 4 + 4096;
    if (hipMalloc(&nb, c) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (*p && keep && (hipMemcpyAsync(nb, *p, keep, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) { (void)hipFree(nb); return -1; }
    if (*p) (void)hipFree(*p);
    *p = nb; *cap = c; return 0;
  }
  // [r5] BGZF streams: the member table reaches `upto`, or the stream's end (then vsize is its size).  A member names its compressed size in its header and
  // its text size in its trailer: the scan hops from member to member and inflates nothing — but the last member of a file, once, for its last byte.
  bool bz_scan(Stream& S, uint64_t upto, std::string* e) {
    using sqio::BgzfSource;
    while (!S.scan_done && S.scan_voff < upto) {
      if (S.scan_file == S.sfiles.size()) { S.scan_done = true; S.final_ = true; S.vsize = S.scan_voff; break; }
      SeqFile& F = *S.sfiles[S.scan_file]; const uint8_t* base = (const uint8_t*)F.map->p; const size_t n = F.map->n;
      if (S.scan_off >= n) {   // the end of a file: a last line without its newline gets one
        if (S.scan_file_bytes) {
          size_t j = S.mem.size(); while (j > 0 && (S.mem[j - 1].file != (uint32_t)S.scan_file || !S.mem[j - 1].isize)) --j;
          if (j > 0) { const Stream::BzMember& M = S.mem[j - 1]; std::vector<char> tmp((size_t)M.isize + 64);
            const char* w = BgzfSource::inflate_member(base + M.coff, BgzfSource::Mem{(size_t)M.coff, (size_t)M.csize, M.isize, 0}, tmp.data());
            if (*w) { *e = "'" + F.path + "': " + w; return false; }
            if (tmp[M.isize - 1] != '\n') { S.mem.push_back(Stream::BzMember{~0u, 0, 0, 1, S.scan_voff, 0, 0}); S.scan_voff += 1; } }
        }
        ++S.scan_file; S.scan_off = 0; S.scan_file_bytes = 0; continue;
      }
      const size_t ms = BgzfSource::member_size(base + S.scan_off, n - S.scan_off);
      if (ms == 0 || ms > n - S.scan_off || ms < 26) { *e = "'" + F.path + "': " + (ms ? "truncated BGZF member" : "not a BGZF member (mixed gzip file?)"); return false; }
      const uint32_t isize = BgzfSource::le32(base + S.scan_off + ms - 4);
      if (isize > (1u << 16)) { *e = "'" + F.path + "': BGZF member larger than 64 KB"; return false; }
      if (isize) { const uint8_t* hp = base + S.scan_off; const uint32_t hdr = 12u + ((uint32_t)hp[10] | ((uint32_t)hp[11] << 8));
        // a member that has text has a deflate stream between its header and its 8-byte trailer: XLEN comes from the file and must leave room for one (the host
        // inflater refuses the same member, BgzfSource::inflate_member; unchecked, the device's stream length `ms - hdr - 8` would wrap)
        if ((size_t)hdr + 8 >= ms) { *e = "'" + F.path + "': truncated BGZF member"; return false; }
        S.mem.push_back(Stream::BzMember{(uint32_t)S.scan_file, (uint32_t)ms, S.scan_off, isize, S.scan_voff, BgzfSource::le32(hp + ms - 8), hdr}); S.scan_voff += isize; S.scan_file_bytes += isize; }
      S.scan_off += ms;
    }
    return true;
  }
  // members [m0, m1) of the stream into dst, the first one from byte `skip` of its text.  The inflater may write up to 32 bytes past a member's text: inside a
  // run that is the next member's place (inflated afterwards), but the last member of the run — the neighbour is another task's — and a member entered
  // in its middle go through a buffer of the thread's own
  bool bz_fill(const Stream& S, size_t m0, size_t m1, uint32_t skip, char* dst, std::string* e) const {
    using sqio::BgzfSource;
    static thread_local std::vector<char> tmp((1u << 16) + 64);
    for (size_t j = m0; j < m1; ++j) {
      const Stream::BzMember& M = S.mem[j]; const uint32_t sk = j == m0 ? skip : 0;
      if (M.file == ~0u) { *dst++ = '\n'; continue; }
      const SeqFile& F = *S.sfiles[M.file]; const uint8_t* base = (const uint8_t*)F.map->p;
      const bool direct = sk == 0 && j + 1 < m1;
      const char* w = BgzfSource::inflate_member(base + M.coff, BgzfSource::Mem{(size_t)M.coff, (size_t)M.csize, M.isize, 0}, direct ? dst : tmp.data());
      if (*w) { *e = "'" + F.path + "': " + w; return false; }
      if (!direct) memcpy(dst, tmp.data() + sk, M.isize - sk);
      dst += M.isize - sk;
    }
    return true;
  }
  // [r5] sequential streams: the text up to `upto` is in the window, or the stream is at its end (then vsize is its size)
  bool seq_ensure(Stream& S, uint64_t upto, std::string* e) {
    static const char nl_pad[2] = "\n";
    while (!S.final_ && S.wend < upto) {
      if (S.sfile_cur == S.sfiles.size()) { S.final_ = true; S.vsize = S.wend; break; }
      PgzBuf B; const int rc = S.sfiles[S.sfile_cur]->next(&B, e);
      if (rc < 0) return false;
      if (rc == 0) {   // the end of a file: one that does not end with a newline gets one, so that records never straddle files
        if (S.file_bytes && S.last_byte != '\n') { PgzBuf P; P.p = nl_pad; P.n = 1; S.win.push_back(SeqChunk{S.wend, P}); S.wend += 1; S.last_byte = '\n'; }
        S.sfiles[S.sfile_cur].reset(); ++S.sfile_cur; S.file_bytes = 0; continue;
      }
      if (!B.n) continue;
      S.last_byte = B.p[B.n - 1]; S.file_bytes += B.n; S.win.push_back(SeqChunk{S.wend, B}); S.wend += B.n;
    }
    return true;
  }
  void seq_trim(Stream& S) { while (!S.win.empty() && S.win.front().voff + S.win.front().b.n <= S.vpos) S.win.pop_front(); }
  bool vread_seq(const Stream& S, uint64_t pos, size_t n, char* dst, std::string* e) {
    size_t lo = 0, hi = S.win.size();   // the last buffer that starts at or before pos
    while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (S.win[mid].voff <= pos) lo = mid; else hi = mid; }
    for (size_t k = lo; n; ++k) {
      if (k >= S.win.size() || S.win[k].voff > pos) { *e = "internal: the reader's text window does not cover a round"; return false; }
      const SeqChunk& c = S.win[k]; const uint64_t in = pos - c.voff; if (in >= c.b.n) continue;
      const size_t take = (size_t)std::min<uint64_t>(c.b.n - in, n); memcpy(dst, c.b.p + in, take); dst += take; pos += take; n -= take;
    }
    return true;
  }
  // bytes [pos, pos + n) of the stream into dst
  bool vread(Stream& S, uint64_t pos, size_t n, char* dst, std::string* e) {
    if (S.seq) return vread_seq(S, pos, n, dst, e);
    size_t k = 0; while (k + 1 < S.files.size() && S.files[k + 1].vbase <= pos) ++k;
    while (n) {
      File& F = S.files[k]; const uint64_t in = pos - F.vbase;
      if (in < F.size) {
        const size_t take = (size_t)std::min<uint64_t>(F.size - in, n); size_t done_b = 0;
        while (done_b < take) { const ssize_t r = pread(F.fd, dst + done_b, take - done_b, (off_t)(in + done_b)); if (r <= 0) { *e = "cannot read '" + F.path + "'"; return false; } done_b += (size_t)r; }
        dst += take; pos += take; n -= take;
      } else if (in < F.size + F.pad) { *dst++ = '\n'; ++pos; --n; }
      else ++k;
    }
    return true;
  }
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void push(Round&& r) { { std::lock_guard<std::mutex> lk(mu); rounds.push_back(std::move(r)); } cv.notify_all(); }
  void give_back(const std::vector<std::pair<int, size_t>>& ps, size_t from = 0) { { std::lock_guard<std::mutex> lk(mu); for (size_t k = from; k < ps.size(); ++k) free_pieces.push_back(ps[k].first); } cv.notify_all(); }
  // the text of up to `want` records of stream i, round by round, to the uploader: *got records in *bytes bytes (whole lines).  false: stop (error in *e, or the reader is closing)
  // hold: the mate's last round is kept back (the caller checks the record counts of the mates before a batch's last round may go out)
  bool stage_text(int i, int si, bool first_of_batch, uint32_t want, bool last_mate, Round* hold, uint32_t* got, size_t* bytes, std::string* e) {
    Stream& S = sm[i]; *got = 0; *bytes = 0;
    const uint64_t need_lines = 4ull * want; uint64_t lines = 0; size_t have = 0; bool reached = false, first = first_of_batch; char last2[2] = {'X', '\n'};
    while (!reached && S.vpos + have < S.vsize) {
      const uint64_t missing = (need_lines - lines + 3) / 4;
      const uint64_t want_more = std::min<uint64_t>((uint64_t)((double)missing * S.est * 1.03) + (1u << 20), (uint64_t)ROUND_PIECES * PIECE);
      if (S.seq) {   // the text this round may take is inflated (bgz: its members are known) — or the end of the stream is — before its pieces are filled
        const double t0 = now();
        if (!(S.bgz ? bz_scan(S, S.vpos + have + want_more + 1, e) : seq_ensure(S, S.vpos + have + want_more + 1, e))) return false;   // + 1: whether the stream ends with this round must be known in this round
        t_wait_text += now() - t0;
        if (S.vpos + have >= S.vsize) break;
      }
      const size_t more = (size_t)std::min<uint64_t>(S.vsize - S.vpos - have, want_more);
      Round r; r.slot = si; r.mate = i; r.dst = have; r.first_of_batch = first; first = false;
      const uint64_t v0 = S.vpos + have;
      // what goes where.  Plain files and buffered streams: pieces of PIECE bytes, a task each.  BGZF: whole members, as many as fit a piece (64 bytes
      // stay free behind them for the inflater), in tasks of about 256 KB of text: a round is ~500 of them, which the pool's threads take as they come
      struct Task { unsigned piece; size_t at; size_t m0, m1; uint32_t skip; };
      std::vector<Task> tasks; std::vector<size_t> fill;
      if (S.bgz) {
        size_t lo = 0, hi = S.mem.size();   // the member that holds v0
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (S.mem[mid].voff <= v0) lo = mid; else hi = mid; }
        size_t mi = lo; uint32_t skip = (uint32_t)(v0 - S.mem[mi].voff); size_t planned = 0;
        while (planned < more && mi < S.mem.size() && fill.size() < (size_t)ROUND_PIECES) {
          size_t f = 0; Task t{(unsigned)fill.size(), 0, mi, mi, skip}; size_t tb = 0;
          while (planned < more && mi < S.mem.size() && f + (S.mem[mi].isize - skip) + 64 <= PIECE) {
            const size_t b = S.mem[mi].isize - skip; f += b; tb += b; planned += b; skip = 0; ++mi; t.m1 = mi;
            if (tb >= (256u << 10)) { tasks.push_back(t); t = Task{(unsigned)fill.size(), f, mi, mi, 0}; tb = 0; }
          }
          if (t.m1 > t.m0) tasks.push_back(t);
          fill.push_back(f);
        }
      } else {
        const unsigned npp = (unsigned)((more + PIECE - 1) / PIECE);
        for (unsigned k = 0; k < npp; ++k) { fill.push_back(std::min(PIECE, more - (size_t)k * PIECE)); tasks.push_back(Task{k, 0, 0, 0, 0}); }
      }
      const unsigned np = (unsigned)fill.size();
      { const double t0 = now(); std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || free_pieces.size() >= np; }); if (stop) return false;
        for (unsigned k = 0; k < np; ++k) { r.pieces.push_back({free_pieces.front(), fill[k]}); free_pieces.pop_front(); } t_wait_piece += now() - t0; }
      std::vector<uint64_t> tcnt(tasks.size(), 0); std::vector<std::string> terrs(tasks.size());
      { const double t0 = now();
      pool->run((unsigned)tasks.size(), [&](unsigned q) {
        const Task& t = tasks[q]; char* dst = ring + (size_t)r.pieces[t.piece].first * PIECE + t.at;
        if (S.bgz) { size_t nb = 0; for (size_t j = t.m0; j < t.m1; ++j) nb += S.mem[j].isize; nb -= t.skip;
          if (!bz_fill(S, t.m0, t.m1, t.skip, dst, &terrs[q])) return;
          tcnt[q] = count_nl(dst, nb); }
        else { if (!vread(S, v0 + (size_t)t.piece * PIECE, r.pieces[t.piece].second, dst, &terrs[q])) return;
          tcnt[q] = count_nl(dst, r.pieces[t.piece].second); }
      });
      t_fill += now() - t0; }
      std::vector<uint64_t> cnt(np, 0); std::vector<std::string> errs(np);
      for (size_t q = 0; q < tasks.size(); ++q) { cnt[tasks[q].piece] += tcnt[q]; if (!terrs[q].empty() && errs[tasks[q].piece].empty()) errs[tasks[q].piece] = terrs[q]; }
      for (unsigned k = 0; k < np; ++k) if (!errs[k].empty()) { *e = errs[k]; give_back(r.pieces); return false; }
      size_t emitted = 0;
      for (unsigned k = 0; k < np; ++k) {
        if (want && lines + cnt[k] >= need_lines) {   // the batch ends in this piece: just behind newline number need_lines
          const char* p0 = ring + (size_t)r.pieces[k].first * PIECE; const char* p = p0; const char* pe = p0 + r.pieces[k].second; uint64_t left = need_lines - lines;
          while (left) { const char* nl = (const char*)memchr(p, '\n', (size_t)(pe - p)); p = nl + 1; --left; }
          r.pieces[k].second = (size_t)(p - p0); emitted += r.pieces[k].second; lines = need_lines; reached = true;
          give_back(r.pieces, k + 1); r.pieces.resize(k + 1);
          break;
        }
        lines += cnt[k]; emitted += r.pieces[k].second;
      }
      const bool at_end = !reached && S.vpos + have + emitted >= S.vsize;
      if (at_end) {   // the end of the input: what is left must be whole records (blank lines at the very end are tolerated, as on the host path)
        // [r5] T = the two bytes in front of the round's tail + the tail: a blank line is a line end that follows a line end, and the one it follows may be the last
        // byte of the previous round or of the previous batch (a batch is cut right behind a newline, so a call starts at the start of a line) — a file whose
        // records fill its batches exactly and then ends with an empty line leaves a round that is nothing but "\n"
        char T[4096 + 2]; size_t tn = 0;   // the last bytes of the round, gathered (they may straddle pieces)
        { size_t want_b = std::min(sizeof(T) - 2, emitted); size_t skip = emitted - want_b;
          for (auto& pc : r.pieces) { if (skip >= pc.second) { skip -= pc.second; continue; } const size_t c = pc.second - skip; memcpy(T + 2 + tn, ring + (size_t)pc.first * PIECE + skip, c); tn += c; skip = 0; } }
        if (tn < emitted) { T[0] = 'X'; T[1] = 'X'; } else { T[0] = last2[0]; T[1] = last2[1]; }
        size_t cut = tn + 2;
        for (;;) {
          if (cut >= 3 && T[cut - 1] == '\n' && T[cut - 2] == '\n') { cut -= 1; --lines; }
          else if (cut >= 4 && T[cut - 1] == '\n' && T[cut - 2] == '\r' && T[cut - 3] == '\n') { cut -= 2; --lines; }
          else break;
        }
        size_t drop = tn + 2 - cut;
        while (drop && !r.pieces.empty()) { auto& pc = r.pieces.back(); const size_t c = std::min(drop, pc.second); pc.second -= c; drop -= c; emitted -= c; if (!pc.second) { give_back(r.pieces, r.pieces.size() - 1); r.pieces.pop_back(); } }
        if (lines % 4) { *e = "'" + S.name + "' ends in the middle of a record (" + std::to_string(lines) + " lines in its last batch)"; give_back(r.pieces); return false; }
      } else if (!reached && emitted) {   // the last two bytes of this round, for the blank-line rule of the round that ends the input
        char b2[2] = {last2[1], 0}; size_t got2 = 0;
        for (size_t q = r.pieces.size(); q-- > 0 && got2 < 2;) { const auto& pc = r.pieces[q]; const char* base = ring + (size_t)pc.first * PIECE; for (size_t c = pc.second; c-- > 0 && got2 < 2;) { b2[1 - got2] = base[c]; ++got2; } }
        if (got2 == 2) { last2[0] = b2[0]; last2[1] = b2[1]; } else if (got2 == 1) { last2[0] = last2[1]; last2[1] = b2[1]; }
      }
      have += emitted;
      if (reached || at_end) {
        *got = reached ? want : (uint32_t)(lines / 4); *bytes = have;
        if (have >= 0xFFFFFFF0ull) { *e = "a batch of " + std::to_string(*got) + " records spans more than 4 GB of text: use a smaller batch"; give_back(r.pieces); return false; }
        r.last_of_mate = true; r.last_of_batch = last_mate; r.n = *got; r.total = have;
        S.vpos = reached ? S.vpos + have : S.vsize;
        if (S.seq) seq_trim(S);
        if (*got) S.est = 0.7 * S.est + 0.3 * ((double)have / (double)*got);
        text_bytes += have;
        if (*got == 0) { give_back(r.pieces); return true; }   // nothing but blank lines was left: no round goes out
      }
      if (hold && r.last_of_mate) *hold = std::move(r); else push(std::move(r));
    }
    return true;
  }
  void produce_stage() {
    (void)hipSetDevice(device);
    auto fail = [&](int rc, const std::string& e) { Round r; r.rc = rc; r.err = e; push(std::move(r)); };
    for (;;) {
      int si = -1;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !free_slots.empty(); }); if (stop) return; si = free_slots.front(); free_slots.pop_front(); }
      const double t0 = now(); std::string e; uint32_t n[2] = {0, 0}; size_t bytes[2] = {0, 0}; Round held;
      bool ok = stage_text(0, si, true, batch, !paired, paired ? nullptr : &held, &n[0], &bytes[0], &e);
      if (ok && paired && n[0]) ok = stage_text(1, si, false, n[0], true, &held, &n[1], &bytes[1], &e);
      if (ok && paired && n[0] && n[1] != n[0]) { ok = false; e = "mate files have different numbers of records (stopped after " + std::to_string(staged_total + n[1]) + " pairs)"; }
      if (ok && paired && !n[0] && sm[1].vpos < sm[1].vsize) {   // the first file is at its end: is there a record left in the second?
        Round probe; uint32_t g = 0; size_t b = 0; ok = stage_text(1, si, true, 1, true, &probe, &g, &b, &e); give_back(probe.pieces);
        if (ok && g) { ok = false; e = "mate files have different numbers of records (stopped after " + std::to_string(staged_total) + " pairs)"; }
      }
      { std::lock_guard<std::mutex> lk(mu); t_stage += now() - t0; if (stop) { for (auto& pc : held.pieces) free_pieces.push_back(pc.first); return; } }
      if (!ok) { give_back(held.pieces); fail(e.empty() ? SQ_ERR_STATE : SQ_ERR_IO, e); return; }
      if (n[0] == 0) { Round r; r.end_of_input = true; { std::lock_guard<std::mutex> lk(mu); free_slots.push_back(si); } push(std::move(r)); return; }
      push(std::move(held));
      staged_total += n[0];
    }
  }
  // one round: its pieces behind each other into the mate's text; with the mate's last round, its splitting kernels; with the batch's, the rest
  int upload(const Round& r, std::string* e) {
    Slot& s = slots[(size_t)r.slot]; hipStream_t st = s.st; const int i = r.mate;
    const int ns = paired ? 2 : 1; const uint32_t stride = (uint32_t)ns;
    if (r.first_of_batch) {
      if (!s.d_err && hipMalloc((void**)&s.d_err, 64) != hipSuccess) { *e = "device allocation failed (reader)"; return SQ_ERR_NOMEM; }
      if (!s.h_res && hipHostMalloc((void**)&s.h_res, 256, hipHostMallocDefault) != hipSuccess) { *e = "page-locked allocation failed (reader)"; return SQ_ERR_NOMEM; }
      if (hipMemsetAsync(s.d_err, 0xFF, 4, st) != hipSuccess || hipMemsetAsync(s.d_err + 1, 0, 12, st) != hipSuccess) { *e = "device failure in the reader"; return SQ_ERR_DEVICE; }
    }
    size_t rb = 0; for (auto& pc : r.pieces) rb += pc.second;
    if (r.dst == 0) { const size_t guess = (size_t)((double)batch * sm[i].est * 1.05) + (8u << 20); if (dev_grow(&s.d_text[i], &s.text_cap[i], std::max(guess, rb + 64))) { *e = "device allocation failed (reader text)"; return SQ_ERR_NOMEM; } }
    if (dev_grow_keep(&s.d_text[i], &s.text_cap[i], r.dst + rb + 64, r.dst, st)) { *e = "device allocation failed (reader text)"; return SQ_ERR_NOMEM; }
    size_t at = r.dst;
    for (auto& pc : r.pieces) { if (hipMemcpyAsync((char*)s.d_text[i] + at, ring + (size_t)pc.first * PIECE, pc.second, hipMemcpyHostToDevice, st) != hipSuccess) { *e = "device copy failed (reader text)"; return SQ_ERR_DEVICE; } at += pc.second; }
    if (r.last_of_mate) {
      const size_t bytes = r.total; s.bytes[i] = bytes; s.n = r.n; const uint32_t nrec = s.n * stride;
      const size_t padded = (bytes + 15) & ~(size_t)15;
      if (hipMemsetAsync((char*)s.d_text[i] + bytes, 0, padded - bytes + 16, st) != hipSuccess) { *e = "device failure in the reader"; return SQ_ERR_DEVICE; }
      const uint64_t nvec = padded / 16; const uint32_t ntile = (uint32_t)((nvec + FQ_TB - 1) / FQ_TB);
      if (dev_grow(&s.d_len, &s.len_cap, ((size_t)nrec + 8) * 4) || dev_grow(&s.d_off, &s.off_cap, ((size_t)nrec + 8) * 8) ||
          dev_grow(&s.d_nlpos[i], &s.nl_cap[i], ((size_t)4 * s.n + 8) * 4) || dev_grow(&s.d_start[i], &s.start_cap[i], ((size_t)s.n + 8) * 4) ||
          dev_grow(&s.d_tile, &s.tile_cap, ((size_t)ntile + 8) * 4) || dev_grow(&s.d_tbase, &s.tbase_cap, ((size_t)ntile + 8) * 8) ||
          dev_grow(&s.d_spine, &s.spine_cap, ((size_t)std::max(sqk::scan_tiles(ntile), sqk::scan_tiles(nrec)) + 8) * 8)) { *e = "device allocation failed (reader)"; return SQ_ERR_NOMEM; }
      k_fq_count<<<ntile, FQ_TB, 0, st>>>((const uint4*)s.d_text[i], nvec, (uint32_t*)s.d_tile);
      sqk::exclusive_scan_u32_u64((const uint32_t*)s.d_tile, (uint64_t*)s.d_tbase, ntile, (uint64_t*)s.d_spine, st);
      k_fq_index<<<ntile, FQ_TB, 0, st>>>((const uint4*)s.d_text[i], nvec, (const uint64_t*)s.d_tbase, (uint32_t*)s.d_nlpos[i], (uint64_t)4 * s.n);
      k_fq_records<<<(s.n + 255) / 256, 256, 0, st>>>((const uint8_t*)s.d_text[i], (const uint32_t*)s.d_nlpos[i], s.n, (uint32_t)i, stride, (uint32_t*)s.d_len, (uint32_t*)s.d_start[i], s.d_err);
      if (r.last_of_batch) {
        // (the sequence buffer is sized by the text, which is more than twice the sequence bytes: a record is '@' + name + sequence + '+' + a quality
        // string as long as the sequence + 4 line ends; a damaged record is reported, and nothing is copied for its batch)
        if (dev_grow(&s.d_seq, &s.seq_cap, (s.bytes[0] + (paired ? s.bytes[1] : 0)) / 2 + 64)) { *e = "device allocation failed (reader sequences)"; return SQ_ERR_NOMEM; }
        sqk::exclusive_scan_u32_u64((const uint32_t*)s.d_len, (uint64_t*)s.d_off, nrec, (uint64_t*)s.d_spine, st);
        for (int m = 0; m < ns; ++m) k_fq_copy<<<(uint32_t)(((uint64_t)s.n * 8 + 255) / 256), 256, 0, st>>>((const uint8_t*)s.d_text[m], (const uint32_t*)s.d_start[m], (const uint64_t*)s.d_off, s.n, (uint32_t)m, stride, (uint8_t*)s.d_seq, s.d_err);
        k_fq_pad<<<1, 64, 0, st>>>((const uint64_t*)s.d_off, nrec, (uint8_t*)s.d_seq, s.d_err);
        if (hipMemcpyAsync(s.h_res, s.d_err, 16, hipMemcpyDeviceToHost, st) != hipSuccess) { *e = "device failure in the reader"; return SQ_ERR_DEVICE; }
      }
    }
    // the pieces go back to the stager once their copies are through (a mate's kernels and a batch's last kernels are waited for here too: 1 of ~20 ms)
    if (hipStreamSynchronize(st) != hipSuccess) { *e = std::string("device failure in the reader: ") + hipGetErrorString(hipGetLastError()); return SQ_ERR_DEVICE; }
    if (r.last_of_batch) {
      const unsigned* herr = s.h_res;
      if (herr[0] != 0xFFFFFFFFu) {
        const unsigned what = herr[1] ? herr[1] : herr[2];
        *e = "record " + std::to_string(total + herr[0]) + (what == 1 ? " does not start with '@'" : what == 2 ? " has no '+' line after one sequence line" : " has a quality string whose length differs from its sequence's") +
             " (multi-line FASTQ? set SQ_READER_DEVICE=0)"; return SQ_ERR_IO; }
    }
    return SQ_OK;
  }
  void produce_upload() {
    (void)hipSetDevice(device);
    for (;;) {
      Round r;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !rounds.empty(); }); if (stop) return; r = std::move(rounds.front()); rounds.pop_front(); }
      if (r.rc != SQ_OK || r.end_of_input) { std::lock_guard<std::mutex> lk(mu); if (r.rc != SQ_OK) { err = r.err; err_rc = r.rc; } done = true; cv.notify_all(); return; }
      std::string e; const double t0 = now(); const int rc = upload(r, &e); const double dt = now() - t0;
      give_back(r.pieces);
      std::lock_guard<std::mutex> lk(mu); t_upload += dt;
      if (rc != SQ_OK) { err = e; err_rc = rc; done = true; stop = true; cv.notify_all(); return; }
      if (r.last_of_batch) { total += slots[(size_t)r.slot].n; ready.push_back(r.slot); cv.notify_all(); }
    }
  }
  // ---- [r5] BGZF inflated on the device ---------------------------------------------------------------------------------------------------------
  void dv_fail(int rc, const std::string& e) { std::lock_guard<std::mutex> lk(mu); if (err_rc == SQ_OK) { err = e; err_rc = rc; } done = true; stop = true; cv.notify_all(); }
  std::deque<std::pair<hipEvent_t, std::vector<int>>> dv_lent;   // ring pieces whose copy to the device may still run, oldest first (stager only)
  // ---- [r6] ordinary gzip files inflated on the device (hip/gzip_dev.hip) ------------------------------------------------------------------------
  // the next chunk of stream i: the next segment of its current file, decoded into the chunk's buffer.  Every stream has a stager thread of its own: a segment is a chain of
  // dependent launches with the host in between (block search -> spans -> windows -> text), and two files' chains fill each other's gaps
  bool gz_make_chunk(int i) {
    Stream& S = sm[i]; DvStream& D = dv[i]; const double t0 = now();
    const int idx = (int)(D.produced % DV_CHUNKS); hipStream_t hs = D.hs[0];
    DvChunk ch; ch.idx = idx; ch.voff = D.ahead; ch.n = 0; ch.last = false; bool have = false, recorded = false;
    while (!have) {
      if (!D.gz) {
        if (D.gz_file == S.sfiles.size()) { ch.last = true; have = true; break; }      // the end of the stream: an empty last chunk
        SeqFile& F = *S.sfiles[D.gz_file]; std::string w;
        const int rc = sq_gzdev_open((const uint8_t*)F.map->p, F.map->n, device, hs, 0, &D.gz, &w);
        if (rc) { dv_fail(rc, "'" + F.path + "': " + w); return false; }
        D.gz_file_bytes = 0; D.gz_last = '\n';
      }
      SeqFile& F = *S.sfiles[D.gz_file]; size_t n = 0; std::string w;
      int rc = sq_gzdev_next(D.gz, &n, &w);
      if (rc) { dv_fail(rc, "'" + F.path + "': " + w); return false; }
      if (n == 0) {      // the file's end (everything queued for it is complete and checked): a last line without its newline gets one
        if (D.gz_file_bytes && D.gz_last_at && hipMemcpy(&D.gz_last, D.gz_last_at, 1, hipMemcpyDeviceToHost) != hipSuccess) { dv_fail(SQ_ERR_DEVICE, "device failure in the reader (gzip)"); return false; }
        sq_gzdev_close(D.gz); D.gz = nullptr; ++D.gz_file; D.gz_last_at = nullptr;
        if (D.gz_file_bytes && D.gz_last != '\n') {
          if (dev_grow(&D.text[idx], &D.text_cap[idx], 64) || hipMemsetAsync(D.text[idx], '\n', 1, hs) != hipSuccess) { dv_fail(SQ_ERR_NOMEM, "device allocation failed (reader: inflated text)"); return false; }
          ch.n = 1; have = true;
        }
        continue;
      }
      if (dev_grow(&D.text[idx], &D.text_cap[idx], n + 64)) { dv_fail(SQ_ERR_NOMEM, "device allocation failed (reader: inflated text)"); return false; }
      rc = sq_gzdev_emit(D.gz, (uint8_t*)D.text[idx], D.ev_done[idx], &w);      // queued on the decoder's own stream: the event is recorded behind the text
      if (rc) { dv_fail(rc, "'" + F.path + "': " + w); return false; }
      D.gz_last_at = (const char*)D.text[idx] + n - 1; D.gz_file_bytes += n; ch.n = n; have = true; recorded = true;
    }
    if (!recorded && hipEventRecord(D.ev_done[idx], hs) != hipSuccess) { dv_fail(SQ_ERR_DEVICE, "device failure in the reader (gzip)"); return false; }      // (the newline chunk, the empty last chunk)
    { std::lock_guard<std::mutex> lk(mu); const bool last = ch.last; const size_t n = ch.n; D.q.push_back(std::move(ch)); D.ahead += n; ++D.produced; if (last) D.finished = true; t_dv_fill += now() - t0; }
    cv.notify_all();
    return true;
  }
  void gz_stream_loop(int i) {
    (void)hipSetDevice(device);
    DvStream& D = dv[i];
    for (;;) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || D.finished || D.q.size() < (size_t)DV_CHUNKS; }); if (stop || D.finished) return; }
      if (!gz_make_chunk(i)) return;
    }
  }
  // the stager: chunk after chunk of the stream that is the least ahead of its splitter
  void produce_inflate() {
    (void)hipSetDevice(device);
    const int ns = paired ? 2 : 1;
    if (any_gzdev) { std::thread second; if (ns == 2) second = std::thread([this] { gz_stream_loop(1); }); gz_stream_loop(0); if (second.joinable()) second.join(); return; }
    for (;;) {
      int i = -1;
      { std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { if (stop) return true; bool all = true; for (int k = 0; k < ns; ++k) { if (!dv[k].finished && dv[k].q.size() < (size_t)DV_CHUNKS) return true; all = all && dv[k].finished; } return all; });
        if (stop) return;
        uint64_t best = ~0ull;
        for (int k = 0; k < ns; ++k) if (!dv[k].finished && dv[k].q.size() < (size_t)DV_CHUNKS) { const uint64_t a = dv[k].ahead - dv[k].pos; if (a < best) { best = a; i = k; } }
        if (i < 0) return; }   // every stream has been inflated to its end
      Stream& S = sm[i]; DvStream& D = dv[i]; std::string e; const double t0 = now();
      const int idx = (int)(D.produced % DV_CHUNKS); hipStream_t hs = D.hs[D.produced & 1];
      if (!bz_scan(S, D.ahead + DV_TEXT + 65536 + 1, &e)) { dv_fail(SQ_ERR_IO, e); return; }
      const size_t m0 = D.next_mem; size_t m1 = m0, tbytes = 0, cbytes = 0;
      while (m1 < S.mem.size() && m1 - m0 < DV_MEMBERS && tbytes < DV_TEXT) { tbytes += S.mem[m1].isize; cbytes += S.mem[m1].csize; ++m1; }
      const uint32_t nmem = (uint32_t)(m1 - m0); const bool last = m1 == S.mem.size() && S.scan_done;
      DvChunk ch; ch.idx = idx; ch.voff = D.ahead; ch.n = tbytes; ch.last = last; ch.where.reserve(nmem);
      if (nmem) {
        // the compressed stream of the chunk = its members' bytes behind each other (headers and trailers included: the descriptors point behind the headers);
        // then the descriptors, in a piece of their own
        std::vector<uint64_t> coffs((size_t)nmem + 1, 0); for (uint32_t k = 0; k < nmem; ++k) coffs[k + 1] = coffs[k] + S.mem[m0 + k].csize;
        const size_t ncp = (cbytes + PIECE - 1) / PIECE, np = ncp + 1;
        if ((size_t)nmem * sizeof(sq_bgzf_member) > PIECE || np > (size_t)RING_PIECES) { dv_fail(SQ_ERR_STATE, "internal: a chunk of BGZF members does not fit the reader's ring"); return; }
        std::vector<int> pcs;
        { std::unique_lock<std::mutex> lk(mu);
          while (free_pieces.size() < np) {
            if (dv_lent.empty()) { lk.unlock(); dv_fail(SQ_ERR_STATE, "internal: the reader's ring ran out of pieces"); return; }
            auto L = std::move(dv_lent.front()); dv_lent.pop_front(); lk.unlock();
            if (hipEventSynchronize(L.first) != hipSuccess) { dv_fail(SQ_ERR_DEVICE, std::string("device failure in the reader: ") + hipGetErrorString(hipGetLastError())); return; }
            lk.lock(); for (int pc : L.second) free_pieces.push_back(pc);
          }
          for (size_t k = 0; k < np; ++k) { pcs.push_back(free_pieces.front()); free_pieces.pop_front(); } }
        constexpr size_t TASK = 1u << 20; const unsigned ntask = (unsigned)((cbytes + TASK - 1) / TASK);
        pool->run(ntask, [&](unsigned t) {
          const uint64_t a = (uint64_t)t * TASK, b = std::min<uint64_t>(a + TASK, cbytes);
          size_t k = (size_t)(std::upper_bound(coffs.begin(), coffs.end(), a) - coffs.begin()) - 1;   // the member that holds byte a
          for (uint64_t at = a; at < b; ++k) {
            const Stream::BzMember& M = S.mem[m0 + k]; if (!M.csize) continue;
            const uint64_t in = at - coffs[k], take = std::min<uint64_t>(M.csize - in, b - at);
            memcpy(ring + (size_t)pcs[(size_t)(at / PIECE)] * PIECE + (size_t)(at % PIECE), (const uint8_t*)S.sfiles[M.file]->map->p + M.coff + in, (size_t)std::min<uint64_t>(take, PIECE - at % PIECE));
            if (take > PIECE - at % PIECE) { const uint64_t d1 = PIECE - at % PIECE; memcpy(ring + (size_t)pcs[(size_t)(at / PIECE) + 1] * PIECE, (const uint8_t*)S.sfiles[M.file]->map->p + M.coff + in + d1, (size_t)(take - d1)); }
            at += take;
          }
        });
        sq_bgzf_member* desc = (sq_bgzf_member*)(ring + (size_t)pcs[ncp] * PIECE); uint64_t tv = 0;
        for (uint32_t k = 0; k < nmem; ++k) { const Stream::BzMember& M = S.mem[m0 + k];
          desc[k] = M.file == ~0u ? sq_bgzf_member{0, tv, 0, M.isize, 0, SQ_BGZF_LINE_END} : sq_bgzf_member{coffs[k] + M.hdr, tv, M.csize - M.hdr - 8, M.isize, M.crc, 0};
          tv += M.isize; ch.where.push_back({M.file, M.coff}); }
        if (dev_grow(&D.text[idx], &D.text_cap[idx], tbytes + 64) || dev_grow(&D.comp[idx], &D.comp_cap[idx], cbytes + 64) || dev_grow(&D.mem[idx], &D.mem_cap[idx], (size_t)nmem * sizeof(sq_bgzf_member) + 64)) {
          dv_fail(SQ_ERR_NOMEM, "device allocation failed (reader: inflated text)"); return; }
        bool okc = true;
        for (size_t k = 0; k < ncp && okc; ++k) okc = hipMemcpyAsync((char*)D.comp[idx] + k * PIECE, ring + (size_t)pcs[k] * PIECE, std::min(PIECE, cbytes - k * PIECE), hipMemcpyHostToDevice, hs) == hipSuccess;
        okc = okc && hipMemcpyAsync(D.mem[idx], desc, (size_t)nmem * sizeof(sq_bgzf_member), hipMemcpyHostToDevice, hs) == hipSuccess && hipEventRecord(D.ev_h2d[idx], hs) == hipSuccess;
        okc = okc && hipMemsetAsync(D.st + 2 * idx, 0xFF, 8, hs) == hipSuccess;
        okc = okc && sq_bgzf_inflate_launch((const uint8_t*)D.comp[idx], (const sq_bgzf_member*)D.mem[idx], nmem, (uint8_t*)D.text[idx], D.st + 2 * idx, hs) == SQ_OK;
        if (!okc) { dv_fail(SQ_ERR_DEVICE, std::string("device failure in the reader (inflate): ") + hipGetErrorString(hipGetLastError())); return; }
        dv_lent.push_back({D.ev_h2d[idx], std::move(pcs)});
      }
      if (hipEventRecord(D.ev_done[idx], hs) != hipSuccess) { dv_fail(SQ_ERR_DEVICE, "device failure in the reader (inflate)"); return; }
      { std::lock_guard<std::mutex> lk(mu); D.q.push_back(std::move(ch)); D.ahead += tbytes; D.next_mem = m1; ++D.produced; if (last) D.finished = true; t_dv_fill += now() - t0; }
      cv.notify_all();
    }
  }
  // the next `want` records of stream i: their text into the slot, their lines indexed, their sequences located.  false: an error (in *e), or the reader is closing (*e empty)
  bool dv_split(int i, int si, uint32_t want, uint32_t* got, std::string* e, int* rc) {
    Stream& S = sm[i]; DvStream& D = dv[i]; Slot& s = slots[(size_t)si]; hipStream_t st = s.st; *got = 0; *rc = SQ_ERR_DEVICE;
    const int ns = paired ? 2 : 1; const uint32_t stride = (uint32_t)ns;
    auto dev_err = [&]() { *e = std::string("device failure in the reader: ") + hipGetErrorString(hipGetLastError()); return false; };
    size_t copied = 0, need = (size_t)((double)want * S.est * 1.02) + (256u << 10); uint64_t lines = 0; bool reached = false; uint64_t pos;
    unsigned* h = s.h_res + 4;   // [0] the last tile's count / a newline position, [2..3] the last tile's base, [4 + 2 k ..] the status of the k-th chunk looked at
    struct Part { int idx; uint64_t voff; size_t n; };
    for (;;) {
      std::vector<Part> parts; uint64_t avail; bool ended;
      { const double t0 = now(); std::unique_lock<std::mutex> lk(mu); pos = D.pos;
        cv.wait(lk, [&] { return stop || (!D.q.empty() && (D.q.back().last || D.q.back().voff + D.q.back().n >= pos + need || D.q.size() == (size_t)DV_CHUNKS)); });
        if (stop) return false;
        t_dv_wait += now() - t0;
        const uint64_t end = D.q.back().voff + D.q.back().n; ended = D.q.back().last && end <= pos + need; avail = end - pos;
        if (!D.q.back().last && end < pos + need) { *rc = SQ_ERR_STATE;
          *e = "a batch of " + std::to_string(want) + " records of '" + S.name + "' takes more text than the device inflater holds at a time (" + std::to_string(need >> 20) + " MB): use a smaller batch, or SQ_READER_BGZF_DEVICE=0"; return false; }
        for (auto& c : D.q) parts.push_back(Part{c.idx, c.voff, c.n}); }
      const size_t take = (size_t)std::min<uint64_t>(avail, need);
      if (take >= 0xFFFFFFF0ull) { *rc = SQ_ERR_STATE; *e = "a batch of " + std::to_string(want) + " records spans more than 4 GB of text: use a smaller batch"; return false; }
      if (dev_grow_keep(&s.d_text[i], &s.text_cap[i], std::max(take, need) + 64, copied, st)) { *rc = SQ_ERR_NOMEM; *e = "device allocation failed (reader text)"; return false; }
      const uint64_t a = pos + copied, b = pos + take; unsigned np = 0;
      for (auto& pt : parts) {
        const uint64_t lo = std::max(a, pt.voff), hi = std::min<uint64_t>(b, pt.voff + pt.n); if (lo >= hi) continue;
        if (hipStreamWaitEvent(st, D.ev_done[pt.idx], 0) != hipSuccess || hipMemcpyAsync((char*)s.d_text[i] + (lo - pos), (const char*)D.text[pt.idx] + (lo - pt.voff), (size_t)(hi - lo), hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(h + 4 + 2 * np, D.st + 2 * pt.idx, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return dev_err();
        parts[np++] = pt;
      }
      copied = take;
      const size_t padded = (copied + 15) & ~(size_t)15; const uint64_t nvec = padded / 16; const uint32_t ntile = (uint32_t)((nvec + FQ_TB - 1) / FQ_TB);
      if (hipMemsetAsync((char*)s.d_text[i] + copied, 0, padded - copied + 16, st) != hipSuccess) return dev_err();
      if (dev_grow(&s.d_tile, &s.tile_cap, ((size_t)ntile + 8) * 4) || dev_grow(&s.d_tbase, &s.tbase_cap, ((size_t)ntile + 8) * 8) ||
          dev_grow(&s.d_spine, &s.spine_cap, ((size_t)std::max(sqk::scan_tiles(ntile), sqk::scan_tiles((uint64_t)batch * stride)) + 8) * 8)) { *rc = SQ_ERR_NOMEM; *e = "device allocation failed (reader)"; return false; }
      h[0] = 0; h[2] = 0; h[3] = 0;
      if (ntile) {
        k_fq_count<<<ntile, FQ_TB, 0, st>>>((const uint4*)s.d_text[i], nvec, (uint32_t*)s.d_tile);
        sqk::exclusive_scan_u32_u64((const uint32_t*)s.d_tile, (uint64_t*)s.d_tbase, ntile, (uint64_t*)s.d_spine, st);
        if (hipMemcpyAsync(h, (const uint32_t*)s.d_tile + (ntile - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(h + 2, (const uint64_t*)s.d_tbase + (ntile - 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess) return dev_err();
      }
      if (hipStreamSynchronize(st) != hipSuccess) return dev_err();
      for (unsigned k = 0; k < np; ++k) if (h[4 + 2 * k] != 0xFFFFFFFFu) {   // a damaged member: which file, where
        std::lock_guard<std::mutex> lk(mu); std::string where = "'" + S.name + "'";
        for (auto& c : D.q) if (c.idx == parts[k].idx && c.voff == parts[k].voff && h[4 + 2 * k] - 1 < c.where.size()) { const auto& w = c.where[h[4 + 2 * k] - 1]; if (w.first != ~0u) where = "'" + S.sfiles[w.first]->path + "': member at offset " + std::to_string(w.second); }
        *rc = SQ_ERR_IO; *e = where + ": " + sq_bgzf_status_text(h[5 + 2 * k]); return false; }
      uint64_t tb; memcpy(&tb, h + 2, 8); lines = tb + h[0];
      if (lines >= 4ull * want) { reached = true; break; }
      if (ended) break;
      need = copied + (size_t)((double)(want - lines / 4) * S.est * 1.05) + (256u << 10);
    }
    uint32_t n = want; size_t used = copied;
    if (!reached) {   // the end of the stream: what is left must be whole records (blank lines at the very end are tolerated, as on the host path)
      char T[4096 + 2]; const size_t tn = std::min(sizeof(T) - 2, copied);
      if (tn && (hipMemcpyAsync(T + 2, (const char*)s.d_text[i] + copied - tn, tn, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) return dev_err();
      T[0] = 'X'; T[1] = tn == copied ? '\n' : 'X';   // a batch starts behind a line end
      size_t cut = tn + 2;
      for (;;) {
        if (cut >= 3 && T[cut - 1] == '\n' && T[cut - 2] == '\n') { cut -= 1; --lines; }
        else if (cut >= 4 && T[cut - 1] == '\n' && T[cut - 2] == '\r' && T[cut - 3] == '\n') { cut -= 2; --lines; }
        else break;
      }
      used = copied - (tn + 2 - cut);
      if (lines % 4) { *rc = SQ_ERR_IO; *e = "'" + S.name + "' ends in the middle of a record (" + std::to_string(lines) + " lines in its last batch)"; return false; }
      n = (uint32_t)(lines / 4);
    }
    if (n) {
      const uint32_t nrec = std::max(n, i == 0 ? n : s.n) * stride;
      if (dev_grow(&s.d_len, &s.len_cap, ((size_t)nrec + 8) * 4) || dev_grow(&s.d_off, &s.off_cap, ((size_t)nrec + 8) * 8) ||
          dev_grow(&s.d_nlpos[i], &s.nl_cap[i], ((size_t)4 * n + 8) * 4) || dev_grow(&s.d_start[i], &s.start_cap[i], ((size_t)n + 8) * 4)) { *rc = SQ_ERR_NOMEM; *e = "device allocation failed (reader)"; return false; }
      const size_t padded = (copied + 15) & ~(size_t)15; const uint64_t nvec = padded / 16; const uint32_t ntile = (uint32_t)((nvec + FQ_TB - 1) / FQ_TB);
      k_fq_index<<<ntile, FQ_TB, 0, st>>>((const uint4*)s.d_text[i], nvec, (const uint64_t*)s.d_tbase, (uint32_t*)s.d_nlpos[i], (uint64_t)4 * n);
      if (reached) {   // the batch ends behind line 4 n
        if (hipMemcpyAsync(h, (const uint32_t*)s.d_nlpos[i] + ((size_t)4 * n - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return dev_err();
        used = (size_t)h[0] + 1;
      }
      k_fq_records<<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)s.d_text[i], (const uint32_t*)s.d_nlpos[i], n, (uint32_t)i, stride, (uint32_t*)s.d_len, (uint32_t*)s.d_start[i], s.d_err);
      S.est = 0.7 * S.est + 0.3 * ((double)used / (double)n);
    }
    s.bytes[i] = used; text_bytes += used; *got = n;
    { std::lock_guard<std::mutex> lk(mu); D.pos += reached ? used : copied;   // chunks whose text is all behind the position go back to the stager (the last one stays: it says the stream is at its end)
      while (!D.q.empty() && !D.q.front().last && D.q.front().voff + D.q.front().n <= D.pos) D.q.pop_front(); }
    cv.notify_all();
    return true;
  }
  // [r5] The usual batch without a wait per step: for BOTH mates the text a full batch is expected to take is copied, its lines counted and indexed, and the three
  // numbers that say whether that was enough (the last tile's count and base, the position of line end 4 * batch) come back in ONE synchronisation.  Anything
  // else — the end of a stream, records longer than expected, a damaged member — leaves the stream positions untouched and goes the careful way (dv_split).
  // 0: issued; 1: not the usual case (the caller takes dv_split); < 0: the reader is closing or a device error (in *e)
  int dv_issue(int i, int si, uint32_t want, unsigned* np_out, std::string* e) {
    Stream& S = sm[i]; DvStream& D = dv[i]; Slot& s = slots[(size_t)si]; hipStream_t st = s.st; const uint32_t stride = paired ? 2u : 1u;
    const size_t need = (size_t)((double)want * S.est * 1.02) + (256u << 10); unsigned* h = s.h_res + 4 + 24 * i;
    struct Part { int idx; uint64_t voff; size_t n; }; std::vector<Part> parts; uint64_t pos;
    { const double t0 = now(); std::unique_lock<std::mutex> lk(mu); pos = D.pos;
      cv.wait(lk, [&] { return stop || (!D.q.empty() && (D.q.back().last || D.q.back().voff + D.q.back().n >= pos + need || D.q.size() == (size_t)DV_CHUNKS)); });
      if (stop) return -1;
      t_dv_wait += now() - t0;
      if (D.q.back().voff + D.q.back().n < pos + need) return 1;      // the stream ends in this batch, or the ring is full
      for (auto& c : D.q) parts.push_back(Part{c.idx, c.voff, c.n}); }
    if (need >= 0xFFFFFFF0ull) return 1;
    const size_t padded = (need + 15) & ~(size_t)15; const uint64_t nvec = padded / 16; const uint32_t ntile = (uint32_t)((nvec + FQ_TB - 1) / FQ_TB);
    // the tile buffers are shared by the mates and mate 1's work may still be queued when mate 2's is issued: they are sized for the larger of the two at once
    size_t need_max = need; for (int k = 0; k < (paired ? 2 : 1); ++k) need_max = std::max(need_max, (size_t)((double)want * sm[k].est * 1.02) + (256u << 10));
    const uint32_t ntile_max = (uint32_t)((((need_max + 15) / 16) + FQ_TB - 1) / FQ_TB) + 1;
    if (dev_grow(&s.d_text[i], &s.text_cap[i], need + 64) || dev_grow(&s.d_tile, &s.tile_cap, ((size_t)ntile_max + 8) * 4) || dev_grow(&s.d_tbase, &s.tbase_cap, ((size_t)ntile_max + 8) * 8) ||
        dev_grow(&s.d_spine, &s.spine_cap, ((size_t)std::max(sqk::scan_tiles(ntile_max), sqk::scan_tiles((uint64_t)batch * stride)) + 8) * 8) ||
        dev_grow(&s.d_len, &s.len_cap, ((size_t)want * stride + 8) * 4) || dev_grow(&s.d_off, &s.off_cap, ((size_t)want * stride + 8) * 8) ||
        dev_grow(&s.d_nlpos[i], &s.nl_cap[i], ((size_t)4 * want + 8) * 4) || dev_grow(&s.d_start[i], &s.start_cap[i], ((size_t)want + 8) * 4)) { *e = "device allocation failed (reader)"; return -2; }
    auto dev_err = [&]() { *e = std::string("device failure in the reader: ") + hipGetErrorString(hipGetLastError()); return -3; };
    unsigned np = 0;
    for (auto& pt : parts) {
      const uint64_t lo = std::max(pos, pt.voff), hi = std::min<uint64_t>(pos + need, pt.voff + pt.n); if (lo >= hi) continue;
      if (hipStreamWaitEvent(st, D.ev_done[pt.idx], 0) != hipSuccess || hipMemcpyAsync((char*)s.d_text[i] + (lo - pos), (const char*)D.text[pt.idx] + (lo - pt.voff), (size_t)(hi - lo), hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpyAsync(h + 4 + 2 * np, D.st + 2 * pt.idx, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return dev_err();
      ++np;
    }
    if (hipMemsetAsync((char*)s.d_text[i] + need, 0, padded - need + 16, st) != hipSuccess) return dev_err();
    k_fq_count<<<ntile, FQ_TB, 0, st>>>((const uint4*)s.d_text[i], nvec, (uint32_t*)s.d_tile);
    // (mate 2 reuses the tile buffers of mate 1: the stream runs mate 1's index before mate 2's count)
    sqk::exclusive_scan_u32_u64((const uint32_t*)s.d_tile, (uint64_t*)s.d_tbase, ntile, (uint64_t*)s.d_spine, st);
    k_fq_index<<<ntile, FQ_TB, 0, st>>>((const uint4*)s.d_text[i], nvec, (const uint64_t*)s.d_tbase, (uint32_t*)s.d_nlpos[i], (uint64_t)4 * want);
    if (hipMemcpyAsync(h, (const uint32_t*)s.d_tile + (ntile - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(h + 2, (const uint64_t*)s.d_tbase + (ntile - 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(h + 1, (const uint32_t*)s.d_nlpos[i] + ((size_t)4 * want - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess) return dev_err();
    *np_out = np; return 0;
  }
  // after the synchronisation: were there 4 * want lines, and every member sound?  Then the batch's text ends behind line 4 * want and the stream moves on
  bool dv_commit(int i, int si, uint32_t want, unsigned np) {
    Stream& S = sm[i]; DvStream& D = dv[i]; Slot& s = slots[(size_t)si]; const unsigned* h = s.h_res + 4 + 24 * i; const uint32_t stride = paired ? 2u : 1u;
    for (unsigned k = 0; k < np; ++k) if (h[4 + 2 * k] != 0xFFFFFFFFu) return false;      // (dv_split names the member)
    uint64_t tb; memcpy(&tb, h + 2, 8); if (tb + h[0] < 4ull * want) return false;
    const size_t used = (size_t)h[1] + 1;
    k_fq_records<<<(want + 255) / 256, 256, 0, s.st>>>((const uint8_t*)s.d_text[i], (const uint32_t*)s.d_nlpos[i], want, (uint32_t)i, stride, (uint32_t*)s.d_len, (uint32_t*)s.d_start[i], s.d_err);
    S.est = 0.7 * S.est + 0.3 * ((double)used / (double)want); s.bytes[i] = used; text_bytes += used;
    { std::lock_guard<std::mutex> lk(mu); D.pos += used; while (!D.q.empty() && !D.q.front().last && D.q.front().voff + D.q.front().n <= D.pos) D.q.pop_front(); }
    cv.notify_all(); return true;
  }
  void produce_split() {
    (void)hipSetDevice(device);
    const int ns = paired ? 2 : 1; const uint32_t stride = (uint32_t)ns;
    for (;;) {
      int si = -1;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !free_slots.empty(); }); if (stop) return; si = free_slots.front(); free_slots.pop_front(); }
      Slot& s = slots[(size_t)si]; hipStream_t st = s.st; const double t0 = now(); std::string e; int rc = SQ_ERR_DEVICE; uint32_t n0 = 0, n1 = 0;
      bool ok = true;
      if (!s.d_err && hipMalloc((void**)&s.d_err, 64) != hipSuccess) { ok = false; rc = SQ_ERR_NOMEM; e = "device allocation failed (reader)"; }
      if (ok && !s.h_res && hipHostMalloc((void**)&s.h_res, 256, hipHostMallocDefault) != hipSuccess) { ok = false; rc = SQ_ERR_NOMEM; e = "page-locked allocation failed (reader)"; }
      if (ok && (hipMemsetAsync(s.d_err, 0xFF, 4, st) != hipSuccess || hipMemsetAsync(s.d_err + 1, 0, 12, st) != hipSuccess)) { ok = false; e = "device failure in the reader"; }
      s.n = 0;
      bool fast = false;
      if (ok) {      // the usual case first: a full batch from every mate, one wait
        unsigned np[2] = {0, 0}; int r0 = dv_issue(0, si, batch, &np[0], &e), r1 = (r0 == 0 && paired) ? dv_issue(1, si, batch, &np[1], &e) : 0;
        if (r0 < 0 || r1 < 0) ok = false;
        else if (r0 == 0 || r1 == 0) {      // something was issued: it has to be through before the buffers are used again, whichever way the batch goes
          if (hipStreamSynchronize(st) != hipSuccess) { ok = false; e = std::string("device failure in the reader: ") + hipGetErrorString(hipGetLastError()); }
          else if (r0 == 0 && r1 == 0) {
            // both mates are looked at before either stream moves: a batch goes the fast way whole or not at all
            auto full = [&](int i) { const unsigned* h = s.h_res + 4 + 24 * i; for (unsigned k = 0; k < np[i]; ++k) if (h[4 + 2 * k] != 0xFFFFFFFFu) return false; uint64_t tb; memcpy(&tb, h + 2, 8); return tb + h[0] >= 4ull * batch; };
            if (full(0) && (!paired || full(1))) { fast = dv_commit(0, si, batch, np[0]) && (!paired || dv_commit(1, si, batch, np[1])); if (fast) { n0 = n1 = batch; s.n = batch; } else { ok = false; rc = SQ_ERR_STATE; e = "internal: the reader's fast path changed its mind"; } }
          }
        }
      }
      if (ok && !fast) ok = dv_split(0, si, batch, &n0, &e, &rc);
      if (ok && !fast) s.n = n0;
      if (ok && !fast && paired && n0) { ok = dv_split(1, si, n0, &n1, &e, &rc);
        if (ok && n1 != n0) { ok = false; rc = SQ_ERR_IO; e = "mate files have different numbers of records (stopped after " + std::to_string(total + n1) + " pairs)"; } }
      if (ok && !fast && paired && !n0) { ok = dv_split(1, si, 1, &n1, &e, &rc);   // the first file is at its end: is there a record left in the second?
        if (ok && n1) { ok = false; rc = SQ_ERR_IO; e = "mate files have different numbers of records (stopped after " + std::to_string(total) + " pairs)"; } }
      if (ok && n0) {
        const uint32_t nrec = n0 * stride;
        if (dev_grow(&s.d_seq, &s.seq_cap, (s.bytes[0] + (paired ? s.bytes[1] : 0)) / 2 + 64)) { ok = false; rc = SQ_ERR_NOMEM; e = "device allocation failed (reader sequences)"; }
        if (ok) {
          sqk::exclusive_scan_u32_u64((const uint32_t*)s.d_len, (uint64_t*)s.d_off, nrec, (uint64_t*)s.d_spine, st);
          for (int m = 0; m < ns; ++m) k_fq_copy<<<(uint32_t)(((uint64_t)n0 * 8 + 255) / 256), 256, 0, st>>>((const uint8_t*)s.d_text[m], (const uint32_t*)s.d_start[m], (const uint64_t*)s.d_off, n0, (uint32_t)m, stride, (uint8_t*)s.d_seq, s.d_err);
          k_fq_pad<<<1, 64, 0, st>>>((const uint64_t*)s.d_off, nrec, (uint8_t*)s.d_seq, s.d_err);
          if (hipMemcpyAsync(s.h_res, s.d_err, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { ok = false; rc = SQ_ERR_DEVICE; e = std::string("device failure in the reader: ") + hipGetErrorString(hipGetLastError()); }
        }
        if (ok && s.h_res[0] != 0xFFFFFFFFu) { const unsigned* herr = s.h_res; const unsigned what = herr[1] ? herr[1] : herr[2]; ok = false; rc = SQ_ERR_IO;
          e = "record " + std::to_string(total + herr[0]) + (what == 1 ? " does not start with '@'" : what == 2 ? " has no '+' line after one sequence line" : " has a quality string whose length differs from its sequence's") + " (multi-line FASTQ? set SQ_READER_DEVICE=0)"; }
      }
      if (!ok) { if (e.empty()) return; dv_fail(rc, e); return; }   // (e empty: the reader is closing)
      std::lock_guard<std::mutex> lk(mu); t_dv_split += now() - t0;
      if (!n0) { free_slots.push_back(si); done = true; cv.notify_all(); return; }
      total += n0; ready.push_back(si); cv.notify_all();
    }
  }
};

int sq_dev_reader_open(const std::vector<std::string>& f1, const std::vector<std::string>& f2, uint32_t batch, uint32_t nslots, sq_dev_reader** out) {
  int dev = 0, ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return SQ_ERR_DEVICE; }   // no device: the caller keeps the host path
  std::unique_ptr<sq_dev_reader> R(new sq_dev_reader()); R->device = dev; R->batch = batch; R->paired = !f2.empty();
  auto close_all = [&]() { for (auto& s : R->sm) for (auto& f : s.files) if (f.fd >= 0) close(f.fd); };
  auto is_gz = [](const std::string& path) { unsigned char mg[2] = {0, 0}; FILE* f = fopen(path.c_str(), "rb"); if (!f) return false; const size_t got = fread(mg, 1, 2, f); fclose(f); return got == 2 && mg[0] == 0x1f && mg[1] == 0x8b; };
  bool any_gz = false, any_plain = false;
  for (int i = 0; i < (R->paired ? 2 : 1); ++i) for (const auto& path : (i ? f2 : f1)) { if (is_gz(path)) any_gz = true; else any_plain = true; }
  if (any_gz && any_plain) return SQ_ERR_DEVICE;   // a mix of compressed and plain files: the host path takes it
  if (any_gz) {   // [r5] compressed input: inflated by a pool of its own into buffers, copied into the ring, split on the device
    const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
    const unsigned nz = getenv("SQ_READER_THREADS") ? (unsigned)atoi(getenv("SQ_READER_THREADS")) : std::min(64u, std::max(2u, hw / 2));   // (128 threads measured no faster than 64 on the GPU box: profiles/r05_reader_compressed.txt)
    R->zpool.reset(new sqio::Pool(std::max(1u, nz)));
    sqio::Pool* zp = R->zpool.get();
    const int nstreams = R->paired ? 2 : 1;
    // [r6] every file an ORDINARY gzip file (none of them BGZF): inflated on the device, hip/gzip_dev.hip (SQ_READER_GZIP_DEVICE=0: by the host's threads, host/pgzip.cpp)
    bool dev_gz = !(getenv("SQ_READER_GZIP_DEVICE") && atoi(getenv("SQ_READER_GZIP_DEVICE")) == 0);
    for (int i = 0; i < nstreams && dev_gz; ++i) for (const auto& path : (i ? f2 : f1)) {
      std::vector<uint8_t> head(70000); FILE* f = fopen(path.c_str(), "rb"); if (!f) { dev_gz = false; break; }
      const size_t got = fread(head.data(), 1, head.size(), f); fclose(f);
      if (got < 18 || sqio::BgzfSource::member_size(head.data(), got)) { dev_gz = false; break; }
    }
    for (int i = 0; i < nstreams; ++i) {
      sq_dev_reader::Stream& S = R->sm[i]; S.seq = true; S.vsize = ~0ull;
      for (const auto& path : (i ? f2 : f1)) {
        std::unique_ptr<sq_dev_reader::SeqFile> F(new sq_dev_reader::SeqFile()); F->path = path;
        int fd = open(path.c_str(), O_RDONLY); struct stat sb;
        if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) { if (fd >= 0) close(fd); sq_set_error("cannot open '%s'", path.c_str()); return SQ_ERR_IO; }
        void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0); close(fd);
        if (m == MAP_FAILED) { sq_set_error("cannot map '%s'", path.c_str()); return SQ_ERR_IO; }
        F->map = std::make_shared<sqio::Mapping>(); F->map->p = m; F->map->n = (size_t)sb.st_size; (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
        if (sqio::BgzfSource::member_size((const uint8_t*)m, (size_t)sb.st_size)) {
          F->is_bgzf = true;
          F->bg.reset(new sqio::BgzfSource()); F->bg->map = F->map; F->bg->base = (const uint8_t*)m; F->bg->n = (size_t)sb.st_size; F->bg->pool = zp; F->bg->path = path;
          F->bg->window = std::max<size_t>(8, (size_t)(2 * nz) / (size_t)nstreams);
        } else if (dev_gz) { S.gzdev = true;      // (nothing to set up here: the stager opens the file's decoder when it gets to it)
        } else {
          const unsigned th = std::max(1u, std::min(32u, nz / (unsigned)nstreams));
          const size_t piece = std::max<size_t>(1u << 20, std::min<size_t>(4u << 20, (size_t)sb.st_size / (4 * th)));
          F->pz = pgz_open((const uint8_t*)m, (size_t)sb.st_size, [zp](std::function<void()> f) { zp->submit(std::move(f)); }, th, piece);
          if (!F->pz) { sq_set_error("'%s' does not start with a gzip member", path.c_str()); return SQ_ERR_IO; }
        }
        S.sfiles.push_back(std::move(F)); S.name = path;
      }
      S.bgz = true; for (auto& F : S.sfiles) if (!F->is_bgzf) S.bgz = false;
      if (S.gzdev) R->any_gzdev = true;
      else if (S.bgz) { for (auto& F : S.sfiles) F->bg.reset(); R->any_bgz = true; }   // the members go straight into the ring (bz_scan / bz_fill): no buffering source
      else R->any_buffered = true;
    }
  } else
  for (int i = 0; i < (R->paired ? 2 : 1); ++i) {
    uint64_t v = 0;
    for (const auto& path : (i ? f2 : f1)) {
      sq_dev_reader::File F; F.path = path; F.fd = open(path.c_str(), O_RDONLY); struct stat sb;
      if (F.fd < 0 || fstat(F.fd, &sb) != 0) { sq_set_error("cannot open '%s'", path.c_str()); close_all(); if (F.fd >= 0) close(F.fd); return SQ_ERR_IO; }
      F.size = (uint64_t)sb.st_size; F.vbase = v;
      if (F.size) { char last = 0; if (pread(F.fd, &last, 1, (off_t)(F.size - 1)) == 1 && last != '\n') F.pad = 1; }
      v += F.size + F.pad; R->sm[i].files.push_back(F); R->sm[i].name = path;
    }
    R->sm[i].vsize = v;
  }
  R->slots.resize(nslots < 2 ? 2 : (nslots > 8 ? 8 : nslots));
  for (auto& s : R->slots) if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError(); for (auto& t : R->slots) if (t.st) (void)hipStreamDestroy(t.st); for (auto& m : R->sm) for (auto& f : m.files) close(f.fd); return SQ_ERR_DEVICE; }
  for (size_t i = 0; i < R->slots.size(); ++i) R->free_slots.push_back((int)i);
  if (!R->any_buffered) R->zpool.reset();
  if (any_gz) { R->RING_PIECES = 96; R->ROUND_PIECES = 32; }
  // [r5] every stream BGZF: the members are inflated on the device (SQ_READER_BGZF_DEVICE=0: by the host's threads, as a mixed or plain-gzip input is)
  R->dev_inflate = R->any_gzdev || (R->any_bgz && !R->any_buffered && !(getenv("SQ_READER_BGZF_DEVICE") && atoi(getenv("SQ_READER_BGZF_DEVICE")) == 0));
  if (R->dev_inflate) {
    auto fail_dv = [&](int rc) { for (auto& D : R->dv) { for (auto& h : D.hs) if (h) (void)hipStreamDestroy(h); for (auto& ev : D.ev_h2d) if (ev) (void)hipEventDestroy(ev); for (auto& ev : D.ev_done) if (ev) (void)hipEventDestroy(ev); if (D.st) (void)hipFree(D.st); }
      for (auto& t : R->slots) if (t.st) (void)hipStreamDestroy(t.st); (void)hipGetLastError(); return rc; };
    double est_max = 0;
    for (int i = 0; i < (R->paired ? 2 : 1); ++i) {
      sq_dev_reader::Stream& S = R->sm[i]; sq_dev_reader::DvStream& D = R->dv[i];
      // the size of a record, from the first member with text (the splitter keeps the estimate current; a first guess far off would make the first batch copy twice)
      if (S.gzdev && !S.sfiles.empty()) {      // [r6] an ordinary gzip file: the first 256 KB of its text through zlib
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 31) == Z_OK) { std::vector<char> tmp(256u << 10); zs.next_in = (Bytef*)const_cast<void*>(S.sfiles[0]->map->p); zs.avail_in = (uInt)std::min<size_t>(S.sfiles[0]->map->n, 1u << 20);
          zs.next_out = (Bytef*)tmp.data(); zs.avail_out = (uInt)tmp.size(); (void)inflate(&zs, Z_SYNC_FLUSH);
          const size_t got = tmp.size() - zs.avail_out; const uint64_t nl = count_nl(tmp.data(), got); if (nl >= 8) S.est = (double)got / ((double)nl / 4.0); inflateEnd(&zs); }
      }
      for (auto& F : S.sfiles) { const uint8_t* base = (const uint8_t*)F->map->p; const size_t n = F->map->n; size_t off = 0; bool found = false;
        while (off < n) { const size_t ms = sqio::BgzfSource::member_size(base + off, n - off); if (ms < 26 || ms > n - off) break;
          const uint32_t isize = sqio::BgzfSource::le32(base + off + ms - 4);
          if (isize && isize <= (1u << 16)) { std::vector<char> tmp((size_t)isize + 64); const char* w = sqio::BgzfSource::inflate_member(base + off, sqio::BgzfSource::Mem{off, ms, isize, 0}, tmp.data());
            if (!*w) { const uint64_t nl = count_nl(tmp.data(), isize); if (nl >= 8) S.est = (double)isize / ((double)nl / 4.0); } found = true; break; }
          off += ms; }
        if (found) break; }
      est_max = std::max(est_max, S.est);
      for (auto& h : D.hs) if (hipStreamCreateWithFlags(&h, hipStreamNonBlocking) != hipSuccess) return fail_dv(SQ_ERR_DEVICE);
      for (int k = 0; k < sq_dev_reader::DV_CHUNKS; ++k) if (hipEventCreateWithFlags(&D.ev_h2d[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&D.ev_done[k], hipEventDisableTiming) != hipSuccess) return fail_dv(SQ_ERR_DEVICE);
      if (hipMalloc((void**)&D.st, sq_dev_reader::DV_CHUNKS * 8) != hipSuccess) return fail_dv(SQ_ERR_NOMEM);
      if (S.gzdev) { uint32_t ok8[2 * sq_dev_reader::DV_CHUNKS]; for (int k = 0; k < sq_dev_reader::DV_CHUNKS; ++k) { ok8[2 * k] = 0xFFFFFFFFu; ok8[2 * k + 1] = 0xFFFFFFFFu; }      // (status words are the BGZF inflater's: "no damaged member")
        if (hipMemcpy(D.st, ok8, sizeof(ok8), hipMemcpyHostToDevice) != hipSuccess) return fail_dv(SQ_ERR_DEVICE); }
    }
    // a chunk is a launch of some thousand waves; the ring of chunk buffers holds at least three batches' text (SQ_READER_BGZF_MEMBERS: fewer members per chunk, for tests)
    R->DV_TEXT = std::max<uint64_t>(256u << 20, (uint64_t)(3.0 * (double)batch * est_max / (double)sq_dev_reader::DV_CHUNKS));
    if (getenv("SQ_READER_BGZF_MEMBERS")) R->DV_MEMBERS = (uint32_t)std::min(32768, std::max(4, atoi(getenv("SQ_READER_BGZF_MEMBERS"))));
    R->RING_PIECES = std::max(96, (int)(2 * (R->DV_TEXT / 2 / sq_dev_reader::PIECE + 2)));   // two chunks' compressed bytes (at the usual ratio and worse) in flight
  }
  const unsigned hw_all = std::max(2u, std::thread::hardware_concurrency());
  // plain files: a few threads move bytes; BGZF: the same pool inflates, so it gets what the buffered streams' pool would have had
  const unsigned nt = getenv("SQ_READER_THREADS") ? (unsigned)atoi(getenv("SQ_READER_THREADS"))
                      : (R->any_bgz && !R->dev_inflate ? std::min(64u, std::max(2u, hw_all / 2)) : std::min(16u, std::max(2u, std::thread::hardware_concurrency() / 4)));
  R->pool.reset(new Workers(std::max(1u, nt)));
  if (hipHostMalloc((void**)&R->ring, (size_t)R->RING_PIECES * sq_dev_reader::PIECE, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError(); R->ring = nullptr; for (auto& t : R->slots) if (t.st) (void)hipStreamDestroy(t.st); for (auto& m : R->sm) for (auto& f : m.files) close(f.fd);
    sq_set_error("page-locked allocation failed (reader: %zu MB)", ((size_t)R->RING_PIECES * sq_dev_reader::PIECE) >> 20); return SQ_ERR_NOMEM; }
  for (int k = 0; k < R->RING_PIECES; ++k) R->free_pieces.push_back(k);
  sq_dev_reader* r = R.release();
  if (r->dev_inflate) { r->prod = std::thread([r] { r->produce_inflate(); }); r->prod2 = std::thread([r] { r->produce_split(); }); }
  else { r->prod = std::thread([r] { r->produce_stage(); }); r->prod2 = std::thread([r] { r->produce_upload(); }); }
  *out = r; return SQ_OK;
}
int sq_dev_reader_next(sq_dev_reader* R, sq_read_batch* b, int* slot) {
  std::unique_lock<std::mutex> lk(R->mu);
  R->cv.wait(lk, [&] { return !R->ready.empty() || R->done; });
  if (R->ready.empty()) { if (R->err_rc != SQ_OK) { sq_set_error("%s", R->err.c_str()); return R->err_rc; } return SQ_OK; }   // b->n == 0: the end
  const int si = R->ready.front(); R->ready.pop_front(); sq_dev_reader::Slot& S = R->slots[(size_t)si];
  b->n = S.n; b->paired = R->paired ? 1 : 0; b->seq = (const uint8_t*)S.d_seq; b->seq_off = (const uint64_t*)S.d_off; b->on_device = 1; *slot = si;
  return SQ_OK;
}
void sq_dev_reader_release(sq_dev_reader* R, int slot) {
  if (slot < 0 || (size_t)slot >= R->slots.size()) return;
  { std::lock_guard<std::mutex> lk(R->mu); R->free_slots.push_back(slot); } R->cv.notify_all();
}
uint64_t sq_dev_reader_total(const sq_dev_reader* R) { return R->total; }
void sq_dev_reader_close(sq_dev_reader* R) {
  if (!R) return;
  { std::lock_guard<std::mutex> lk(R->mu); R->stop = true; } R->cv.notify_all();
  if (R->prod.joinable()) R->prod.join();
  if (R->prod2.joinable()) R->prod2.join();
  if (getenv("SQ_READER_STATS") && !R->dev_inflate) fprintf(stderr, "[sq_dev_reader] %llu records, %.3f GB of text: staging %.3f s (%.1f GB/s; %.3f s of it waiting for ring pieces, %.3f s for inflated text / member scans, %.3f s filling the pieces), upload + split %.3f s (%.1f GB/s)\n", (unsigned long long)R->total,
      (double)R->text_bytes / 1e9, R->t_stage, (double)R->text_bytes / 1e9 / std::max(R->t_stage, 1e-9), R->t_wait_piece, R->t_wait_text, R->t_fill, R->t_upload, (double)R->text_bytes / 1e9 / std::max(R->t_upload, 1e-9));
  if (getenv("SQ_READER_STATS") && R->dev_inflate) fprintf(stderr, "[sq_dev_reader] %llu records, %.3f GB of text, BGZF inflated on the device (chunks of %llu MB of text): stager busy %.3f s, splitter busy %.3f s (%.3f s of it waiting for inflated text)\n",
      (unsigned long long)R->total, (double)R->text_bytes / 1e9, (unsigned long long)(R->DV_TEXT >> 20), R->t_dv_fill, R->t_dv_split, R->t_dv_wait);
  R->pool.reset(); (void)hipSetDevice(R->device);
  for (auto& D : R->dv) {
    for (auto& h : D.hs) if (h) (void)hipStreamSynchronize(h);
    if (D.gz) { sq_gzdev_close(D.gz); D.gz = nullptr; }
    for (auto& h : D.hs) if (h) (void)hipStreamDestroy(h);
    for (auto& ev : D.ev_h2d) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : D.ev_done) if (ev) (void)hipEventDestroy(ev);
    for (int k = 0; k < sq_dev_reader::DV_CHUNKS; ++k) for (void* p : {D.text[k], D.comp[k], D.mem[k]}) if (p) (void)hipFree(p);
    if (D.st) (void)hipFree(D.st);
  }
  for (auto& m : R->sm) { m.win.clear(); m.sfiles.clear(); }   // the sources' tasks run on zpool: they go first
  R->zpool.reset();
  for (auto& s : R->slots) {
    if (s.st) { (void)hipStreamSynchronize(s.st); (void)hipStreamDestroy(s.st); }
    if (s.h_res) (void)hipHostFree(s.h_res);
    for (int i = 0; i < 2; ++i) { if (s.d_text[i]) (void)hipFree(s.d_text[i]); if (s.d_nlpos[i]) (void)hipFree(s.d_nlpos[i]); if (s.d_start[i]) (void)hipFree(s.d_start[i]); }
    for (void* p : {s.d_tile, s.d_tbase, s.d_spine, s.d_len, s.d_off, s.d_seq, (void*)s.d_err}) if (p) (void)hipFree(p);
  }
  if (R->ring) (void)hipHostFree(R->ring);
  for (auto& sm : R->sm) for (auto& f : sm.files) if (f.fd >= 0) close(f.fd);
  delete R;
}
