// tools/gather_bench.hip — random 64-byte-sector gather ceiling on MI355X (SURVEY.md §8d asks for it as
// the honest denominator of the k-mer lookup kernel): every lane walks `steps` loads; with chain=1 the
// address of load i+1 depends on the value of load i (the dictionary's pilot -> slot -> pool chain),
// with chain=0 the addresses are independent (what a perfect prefetcher could reach).
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/_build/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int CHAIN>
__global__ void k_gather(const uint64_t* __restrict__ tab, uint64_t nlines, int steps, uint64_t* __restrict__ out) {
  uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t h = mix(gid + 1), acc = 0;
  for (int i = 0; i < steps; ++i) {
    const uint64_t line = h % nlines;
    const uint64_t v = tab[line * 8 + (h >> 61)];           // one 8-byte word of a random 64-byte line
    acc += v;
    h = CHAIN ? mix(h ^ v) : mix(h + 0x9E3779B97F4A7C15ULL);
  }
  out[gid] = acc;
}
int main(int argc, char** argv) {
  const size_t sizes_mb[] = {8, 64, 232, 1024, 8192};
  uint64_t* out; const int TB = 256; const int steps = 32;
  for (size_t mb : sizes_mb) {
    const uint64_t nlines = mb * 1024 * 1024 / 64; uint64_t* tab;
    if (hipMalloc(&tab, nlines * 64) != hipSuccess) { printf("alloc %zu MB failed\n", mb); continue; }
    hipMemset(tab, 1, nlines * 64);
    for (int occ : {4, 8}) {
      const uint32_t blocks = 256 * occ; const uint64_t nthr = (uint64_t)blocks * TB;
      hipMalloc(&out, nthr * 8);
      for (int chain = 0; chain < 2; ++chain) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (chain) k_gather<1><<<blocks, TB>>>(tab, nlines, steps, out); else k_gather<0><<<blocks, TB>>>(tab, nlines, steps, out);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep == 2) printf("table %5zu MB  blocks/CU %d  chain %d : %.3f ms  %.2f G lines/s  %.1f GB/s (64 B lines)\n", mb, occ, chain, ms, nthr * steps / ms / 1e6, nthr * steps * 64.0 / ms / 1e6);
        }
      }
      hipFree(out);
    }
    hipFree(tab);
  }
  return 0;
}
