// tools/fetch_calib.hip — what rocprofv3's FETCH_SIZE counts for the access patterns of k_seed2, on known byte counts (the guide: FETCH_SIZE reports half the bytes of a
// wide coalesced streaming read on gfx950; other patterns are to be calibrated).  Three kernels over a 4 GB table (far past the 256 MB Infinity Cache), each touching a
// known number of 64-byte lines exactly once per load:
//   k_cal_gather8     a random line per lane and step, ONE 8-byte word of it                      (the minimizer table / string pool / bounds reads)
//   k_cal_gather_lds  a random line per lane and step, all 64 bytes by four 16-byte LDS-DMA loads (the filter block: global_load_lds_dwordx4)
//   k_cal_stream16    consecutive 16 bytes per lane                                               (the guide's case, as a control)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; the program prints the bytes each kernel asked for, tools/pmc_summary.py what the counter saw.
// build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/_build/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void k_cal_gather8(const uint64_t* __restrict__ tab, uint64_t nlines, int steps, uint64_t* __restrict__ out) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; uint64_t h = mix(gid + 1), acc = 0;
  for (int i = 0; i < steps; ++i) { acc += tab[(h % nlines) * 8 + (h >> 61)]; h = mix(h + 0x9E3779B97F4A7C15ULL); }
  out[gid] = acc;
}
__global__ void __launch_bounds__(256) k_cal_gather_lds(const uint64_t* __restrict__ tab, uint64_t nlines, int steps, uint64_t* __restrict__ out) {
  __shared__ uint4 s_q[4][256];      // quarter q of lane t's line at [q][t]: the layout the LDS-DMA writes (wave base + lane x 16 bytes)
  typedef __attribute__((address_space(3))) void* lds_vp; typedef const __attribute__((address_space(1))) void* glb_vp;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; const uint32_t tx = threadIdx.x; uint64_t h = mix(gid + 1), acc = 0;
  for (int i = 0; i < steps; ++i) {
    const char* g = (const char*)(tab + (h % nlines) * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds((glb_vp)(g + 16 * q), (lds_vp)&s_q[q][tx & ~63u], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    acc += s_q[h >> 62][tx].x; h = mix(h + 0x9E3779B97F4A7C15ULL);
  }
  out[gid] = acc;
}
__global__ void k_cal_stream16(const uint4* __restrict__ tab, uint64_t nvec, uint64_t* __restrict__ out) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (uint64_t)gridDim.x * blockDim.x; uint64_t acc = 0;
  for (uint64_t i = gid; i < nvec; i += nthr) { const uint4 v = tab[i]; acc += v.x + v.w; }
  out[gid] = acc;
}
int main() {
  const uint64_t bytes = 4ull << 30, nlines = bytes / 64; uint64_t* tab; uint64_t* out; const int TB = 256, blocks = 256 * 8, steps = 64;
  if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, (size_t)blocks * TB * 8) != hipSuccess) { printf("allocation failed\n"); return 1; }
  hipMemset(tab, 1, bytes); hipDeviceSynchronize();
  const double lines = (double)blocks * TB * steps;
  k_cal_gather8<<<blocks, TB>>>(tab, nlines, steps, out); hipDeviceSynchronize();
  k_cal_gather_lds<<<blocks, TB>>>(tab, nlines, steps, out); hipDeviceSynchronize();
  k_cal_stream16<<<blocks, TB>>>((const uint4*)tab, bytes / 16, out); hipDeviceSynchronize();
  printf("k_cal_gather8     asks for %.0f lines of 64 B = %.0f bytes (%.0f bytes actually used)\n", lines, lines * 64, lines * 8);
  printf("k_cal_gather_lds  asks for %.0f lines of 64 B = %.0f bytes\n", lines, lines * 64);
  printf("k_cal_stream16    asks for %.0f bytes\n", (double)bytes);
  return 0;
}
