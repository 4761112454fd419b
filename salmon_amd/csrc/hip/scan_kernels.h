// scan_kernels.h — exclusive prefix sum u32 -> u64 over a batch (offsets of per-end MEM counts, per-fragment alignment counts,
// assigned-fragment flags).  Three short launches on the caller's stream: per-tile totals, an in-place exclusive scan of the
// tile totals by one block (the "spine": 8 M items are 2048 tiles), then the tiles again with their base added.  The input is
// read twice (8 B/item) and the output written once (8 B/item): 16 B/item, HBM-bound, a few tens of µs per batch.
//   out[i] = in[0] + ... + in[i-1]   for i = 0 .. n   (n + 1 outputs; in[n] is never read)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sqk {

constexpr int SCAN_TB = 512;                          // threads per tile
constexpr int SCAN_IPT = 8;                           // items per thread
constexpr int SCAN_TILE = SCAN_TB * SCAN_IPT;         // 4096 items per tile

__device__ inline uint64_t scan_wave_incl(uint64_t v) {   // inclusive scan across the 64 lanes of a wave
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
// exclusive scan of one value per thread across a block of NT threads (NT a multiple of 64, <= 1024); *total = block sum
template <int NT>
__device__ inline uint64_t scan_block_excl(uint64_t v, uint64_t* total, uint64_t* wsum /* LDS [NT/64 + 1] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t inc = scan_wave_incl(v);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) { uint64_t acc = 0; for (int i = 0; i < NT / 64; ++i) { const uint64_t t = wsum[i]; wsum[i] = acc; acc += t; } wsum[NT / 64] = acc; }
  __syncthreads();
  const uint64_t r = wsum[w] + inc - v;
  *total = wsum[NT / 64];
  __syncthreads();   // wsum may be reused by the caller's next round
  return r;
}

static __global__ __launch_bounds__(SCAN_TB) void k_scan_tile_totals(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ tile_total) {
  __shared__ uint64_t wsum[SCAN_TB / 64 + 1];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {   // strided: consecutive lanes read consecutive words
    const uint64_t i = base + (uint64_t)k * SCAN_TB + threadIdx.x;
    if (i < n) s += in[i];
  }
  uint64_t tot; (void)scan_block_excl<SCAN_TB>(s, &tot, wsum);
  if (threadIdx.x == 0) tile_total[blockIdx.x] = tot;
}
static __global__ __launch_bounds__(1024) void k_scan_spine(uint64_t* __restrict__ tile_total, uint32_t nt) {
  __shared__ uint64_t wsum[1024 / 64 + 1];
  uint64_t carry = 0;
  for (uint32_t b = 0; b < nt; b += 1024) {
    const uint32_t i = b + threadIdx.x;
    const uint64_t v = i < nt ? tile_total[i] : 0;
    uint64_t tot; const uint64_t e = scan_block_excl<1024>(v, &tot, wsum);
    if (i < nt) tile_total[i] = carry + e;
    carry += tot;
  }
}
static __global__ __launch_bounds__(SCAN_TB) void k_scan_tiles(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ tile_base,
    uint64_t* __restrict__ out) {
  __shared__ uint64_t wsum[SCAN_TB / 64 + 1];
  __shared__ uint32_t stage[SCAN_TILE];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
  // coalesced load into LDS, then each thread owns SCAN_IPT consecutive items
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    const uint32_t j = (uint32_t)k * SCAN_TB + threadIdx.x; const uint64_t i = base + j;
    stage[j] = i < n ? in[i] : 0u;
  }
  __syncthreads();
  uint32_t v[SCAN_IPT]; uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) { v[k] = stage[threadIdx.x * SCAN_IPT + k]; s += v[k]; }
  uint64_t tot; uint64_t run = tile_base[blockIdx.x] + scan_block_excl<SCAN_TB>(s, &tot, wsum);
  const uint64_t i0 = base + (uint64_t)threadIdx.x * SCAN_IPT;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) { if (i0 + k <= n) out[i0 + k] = run; run += v[k]; }
}

// enqueue the three launches; `spine` holds one u64 per tile: scan_tiles(n) entries
inline uint32_t scan_tiles(uint64_t n) { return (uint32_t)((n + 1 + SCAN_TILE - 1) / SCAN_TILE); }
inline void exclusive_scan_u32_u64(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* spine, hipStream_t st) {
  const uint32_t nt = scan_tiles(n);
  k_scan_tile_totals<<<nt, SCAN_TB, 0, st>>>(in, n, spine);
  k_scan_spine<<<1, 1024, 0, st>>>(spine, nt);
  k_scan_tiles<<<nt, SCAN_TB, 0, st>>>(in, n, spine, out);
}

}  // namespace sqk
