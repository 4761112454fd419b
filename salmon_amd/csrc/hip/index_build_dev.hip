// hip/index_build_dev.hip — [r6] the k-mer table of index construction on the device ("GPU-assisted sort / partition").
//
// host/index_build.cpp decides where unitigs end from what stands next to every occurrence of every canonical k-mer: a concurrent hash table
// over the k-mers (edge masks + terminal flags), then one look-up per reference position that leaves two bits per nucleotide (`brkR`, `brkL`:
// the unitig ends after / before this occurrence).  On the host that table lives under a memory budget and the references are walked once per
// PARTITION of the k-mers (five times for a 3 Gnt decoy genome), every step a random access into gigabytes: the largest part of a build whose
// threads, on the GPU boxes of this pool, share 16 cores' worth of CPU time.  HBM holds the whole table (12 bytes per slot, 1.35 slots per
// position: 50 GB for 3.1·10^9 positions) and takes random 8-byte atomics at a rate no host does, so here it is ONE partition (more only when
// the device's free memory says so), two launches:
//   k_kt_insert  a thread rolls the k-mers of a run of 32 consecutive positions of a reference, claims each canonical k-mer's slot by compare-and-swap
//                on the key (open addressing, linear probing) and ORs what it saw next to it into the slot's info word;
//   k_kt_breaks  the same walk again: look the k-mer up, apply the predicate (a terminal, or not exactly one neighbour), OR the two bit arrays.
// What comes back — the two bit arrays — is a pure function of the SET of (k-mer, neighbours) observations: it does not depend on the table's
// size, hash, partitioning or insertion order, so the index is byte for byte the host builder's (tests/test_index_device_build.py holds both to
// each other; everything behind this phase is the host's code either way).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "device_index.h"

namespace {
constexpr uint32_t KT_RUN = 32;          // positions per thread
struct KtPiece { uint64_t g; uint32_t nk, i0, n, pad; };   // reference start (global nt), its k-mers, first k-mer of the piece, k-mers in the piece
constexpr uint32_t KT_PIECE = 256 * KT_RUN;   // a block's worth

__device__ inline uint64_t kt_home(uint64_t c, uint64_t cap) { return __umul64hi(sq_mix64(c), cap); }
__device__ inline uint32_t kt_part(uint64_t c, uint32_t nparts) { return nparts == 1 ? 0u : (uint32_t)((sq_mix64(c ^ 0x51ED270B1A2C3D4FULL) >> 17) % nparts); }

template <bool INSERT>
__global__ void __launch_bounds__(256) k_kt_pass(const uint64_t* __restrict__ rs, const KtPiece* __restrict__ pieces, uint32_t npieces, uint32_t k, uint32_t part, uint32_t nparts,
                                                 unsigned long long* __restrict__ keys, uint32_t* __restrict__ info, uint64_t cap,
                                                 unsigned long long* __restrict__ brkR, unsigned long long* __restrict__ brkL, uint32_t* __restrict__ full) {
  const KtPiece pc = pieces[blockIdx.x];
  const uint32_t j0 = threadIdx.x * KT_RUN; if (j0 >= pc.n) return;
  const uint32_t j1 = min(pc.n, j0 + KT_RUN);
  const uint64_t km = sq_kmask(k);
  uint32_t i = pc.i0 + j0;
  uint64_t fw = sq_fetch_bases(rs, pc.g + i, k), rc = sq_revcomp(fw, k);
  for (uint32_t j = j0; j < j1; ++j, ++i) {
    if (j > j0) { const uint64_t nb = sq_fetch_base(rs, pc.g + i + k - 1); fw = (fw >> 2) | (nb << (2 * (k - 1))); rc = ((rc << 2) | (3 - nb)) & km; }
    const bool o1 = fw < rc; const uint64_t c = o1 ? fw : rc;
    if (kt_part(c, nparts) != part) continue;
    uint64_t h = kt_home(c, cap); bool found = false;
    for (uint64_t probes = 0; probes < cap; ++probes) {
      unsigned long long cur = INSERT ? __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : keys[h];
      if (INSERT && cur == ~0ULL) { cur = atomicCAS(&keys[h], ~0ULL, (unsigned long long)c); if (cur == ~0ULL) cur = c; }
      if (cur == c) { found = true; break; }
      if (!INSERT && cur == ~0ULL) break;
      if (++h == cap) h = 0;
    }
    if (!found) { atomicExch(full, 1u); return; }   // INSERT: every slot taken (the host sizes the table again); look-up: cannot happen after a complete insert pass
    if (INSERT) {
      uint32_t bits = 0;   // bits 0-3 R edge mask, 4-7 L edge mask, 8 Rterm, 9 Lterm — in the canonical k-mer's orientation (host/index_build.cpp: KTable::info)
      if (i + 1 < pc.nk) { const uint32_t sb = sq_fetch_base(rs, pc.g + i + k); bits |= o1 ? (1u << sb) : (1u << (4 + (3 - sb))); }
      else bits |= o1 ? (1u << 8) : (1u << 9);
      if (i > 0) { const uint32_t pb = sq_fetch_base(rs, pc.g + i - 1); bits |= o1 ? (1u << (4 + pb)) : (1u << (3 - pb)); }
      else bits |= o1 ? (1u << 9) : (1u << 8);
      atomicOr(&info[h], bits);
    } else {
      const uint32_t inf = info[h]; const uint64_t gp = pc.g + i;
      const uint32_t mR = o1 ? (inf & 15u) : ((inf >> 4) & 15u), mL = o1 ? ((inf >> 4) & 15u) : (inf & 15u);
      const bool tR = o1 ? ((inf >> 8) & 1u) : ((inf >> 9) & 1u), tL = o1 ? ((inf >> 9) & 1u) : ((inf >> 8) & 1u);
      if (tR || __popc(mR) != 1) atomicOr(&brkR[gp >> 6], 1ULL << (gp & 63));   // the unitig ends after this occurrence
      if (tL || __popc(mL) != 1) atomicOr(&brkL[gp >> 6], 1ULL << (gp & 63));   // ... before it
    }
  }
}
struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) (void)hipFree(p); } int get(size_t n) { return hipMalloc(&p, n ? n : 16) == hipSuccess ? 0 : -1; } };
}  // namespace

// brkR / brkL: (total_nt + 63) / 64 + 1 words each, zeroed by the caller.  Returns SQ_OK, SQ_ERR_DEVICE when there is no such device (nothing has been touched: the
// caller may take the host's passes), SQ_ERR_NOMEM when not even a sixteenth of the positions' table fits, or what went wrong on the device.
int sq_index_breaks_dev(int device, const uint64_t* refseq, uint64_t refseq_words, const uint32_t* ref_len, const uint64_t* ref_accum, uint32_t nrefs, uint32_t k,
                        uint64_t total_nt, uint64_t* brkR, uint64_t* brkL) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { (void)hipGetLastError(); sq_set_error("no HIP device %d for the index builder's k-mer table", device); return SQ_ERR_DEVICE; }
  SQ_HIP_CHECK(hipSetDevice(device));
  const bool timing = getenv("SQ_TIMING") != nullptr; const auto t0 = std::chrono::steady_clock::now();
  std::vector<KtPiece> pieces; uint64_t npos = 0;
  for (uint32_t r = 0; r < nrefs; ++r) { const uint32_t L = ref_len[r]; if (L < k) continue; const uint32_t nk = L - k + 1; npos += nk;
    for (uint32_t i0 = 0; i0 < nk; i0 += KT_PIECE) pieces.push_back(KtPiece{ref_accum[r], nk, i0, std::min(KT_PIECE, nk - i0), 0}); }
  const uint64_t bw = (total_nt + 63) / 64 + 1;
  if (pieces.empty()) return SQ_OK;
  if (pieces.size() > 0x7FFFFFFFull) { sq_set_error("index builder: too many reference pieces for one launch"); return SQ_ERR_ARG; }
  DevBuf d_rs, d_pc, d_keys, d_info, d_R, d_L, d_full;
  if (d_rs.get(refseq_words * 8) || d_pc.get(pieces.size() * sizeof(KtPiece)) || d_R.get(bw * 8) || d_L.get(bw * 8) || d_full.get(16)) { sq_set_error("device allocation failed (index builder: references / bit arrays)"); return SQ_ERR_NOMEM; }
  SQ_HIP_CHECK(hipMemcpy(d_rs.p, refseq, refseq_words * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(d_pc.p, pieces.data(), pieces.size() * sizeof(KtPiece), hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemset(d_R.p, 0, bw * 8)); SQ_HIP_CHECK(hipMemset(d_L.p, 0, bw * 8));
  // the table: 1.35 slots per position of a partition, 12 bytes per slot; as few partitions as the free memory allows (SQ_INDEX_DEVICE_GB caps it: tests run several)
  size_t freeb = 0, totb = 0; SQ_HIP_CHECK(hipMemGetInfo(&freeb, &totb));
  double budget = (double)freeb * 0.85; if (getenv("SQ_INDEX_DEVICE_GB")) budget = std::min(budget, std::max(1e-4, atof(getenv("SQ_INDEX_DEVICE_GB"))) * 1e9);
  uint32_t nparts = 1; while (nparts < 16 && (double)npos / nparts * 1.08 * 1.35 * 12.0 + 4096 * 12.0 > budget) ++nparts;
  if ((double)npos / nparts * 1.08 * 1.35 * 12.0 + 4096 * 12.0 > budget) { sq_set_error("index builder: the device's free memory (%.1f GB) does not hold a sixteenth of the k-mer table", (double)freeb / 1e9); return SQ_ERR_NOMEM; }
  double grow = 1.0; uint64_t cap = 0;
  for (uint32_t part = 0; part < nparts; ++part) {
    const uint64_t want = (uint64_t)((double)npos / nparts * (nparts > 1 ? 1.08 : 1.0) * 1.35 * grow) + 4096;
    if (want != cap) { if (d_keys.p) { (void)hipFree(d_keys.p); d_keys.p = nullptr; } if (d_info.p) { (void)hipFree(d_info.p); d_info.p = nullptr; } cap = want;
      if (d_keys.get(cap * 8) || d_info.get(cap * 4)) { sq_set_error("device allocation failed (index builder: k-mer table of %.1f GB)", (double)cap * 12.0 / 1e9); return SQ_ERR_NOMEM; } }
    SQ_HIP_CHECK(hipMemset(d_keys.p, 0xFF, cap * 8)); SQ_HIP_CHECK(hipMemset(d_info.p, 0, cap * 4)); SQ_HIP_CHECK(hipMemset(d_full.p, 0, 4));
    k_kt_pass<true><<<(uint32_t)pieces.size(), 256>>>((const uint64_t*)d_rs.p, (const KtPiece*)d_pc.p, (uint32_t)pieces.size(), k, part, nparts, (unsigned long long*)d_keys.p, (uint32_t*)d_info.p, cap,
                                                      (unsigned long long*)d_R.p, (unsigned long long*)d_L.p, (uint32_t*)d_full.p);
    uint32_t full = 0; SQ_HIP_CHECK(hipMemcpy(&full, d_full.p, 4, hipMemcpyDeviceToHost));
    if (full) {   // a partition with more distinct k-mers than it was sized for (a skewed split under a small budget): again, larger
      if (grow > 64.0) { sq_set_error("index builder: the device k-mer table keeps overflowing"); return SQ_ERR_OVERFLOW; }
      grow *= 1.5; --part; continue;
    }
    k_kt_pass<false><<<(uint32_t)pieces.size(), 256>>>((const uint64_t*)d_rs.p, (const KtPiece*)d_pc.p, (uint32_t)pieces.size(), k, part, nparts, (unsigned long long*)d_keys.p, (uint32_t*)d_info.p, cap,
                                                       (unsigned long long*)d_R.p, (unsigned long long*)d_L.p, (uint32_t*)d_full.p);
    SQ_HIP_CHECK(hipMemcpy(&full, d_full.p, 4, hipMemcpyDeviceToHost));
    if (full) { sq_set_error("internal: the index builder's look-up pass met a k-mer the insert pass had not entered"); return SQ_ERR_STATE; }
  }
  SQ_HIP_CHECK(hipMemcpy(brkR, d_R.p, bw * 8, hipMemcpyDeviceToHost)); SQ_HIP_CHECK(hipMemcpy(brkL, d_L.p, bw * 8, hipMemcpyDeviceToHost));
  if (timing) fprintf(stderr, "[sq-timing] index k-mer table on device %d: %llu positions, %u partition(s) of %.1f GB, %.2f s (copies included)\n", device, (unsigned long long)npos, nparts,
      (double)cap * 12.0 / 1e9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return SQ_OK;
}
