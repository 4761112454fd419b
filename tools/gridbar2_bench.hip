// tools/gridbar2_bench.hip — can a persistent multi-phase kernel on MI355X hand data from phase to phase WITHOUT cache-maintenance fences?
// tools/gridbar_bench.hip measured a grid barrier with __threadfence() at 12-43 us per phase (the fence writes back / invalidates the XCD's L2).
// Here every value one phase hands to the next is written with an agent-scope atomic store and read with an agent-scope atomic load (the
// instructions carry sc1: they are served by the memory side, not by the per-XCD L2), the barrier is a counter + generation word touched only
// with agent-scope atomics, and a thread drains its memory operations (s_waitcnt 0) before its block arrives.  The phase has the shape of an EM
// half-iteration: every one of M outputs sums G/M gathered values of the previous phase's M outputs.  The result is compared bit for bit with
// the same phases run as one launch each (plain loads and stores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
__device__ inline double ld_ag(const double* p) { return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ inline void st_ag(double* p, double v) { __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t mixh(uint32_t a, uint32_t b) { uint32_t x = a * 0x9E3779B1u + b * 0x85EBCA77u; x ^= x >> 15; x *= 0xC2B2AE3Du; x ^= x >> 13; return x; }
__device__ inline void bar(unsigned* b, unsigned nblocks) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gen = __hip_atomic_load(&b[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(&b[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
      __hip_atomic_store(&b[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);
      (void)__hip_atomic_fetch_add(&b[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else while (__hip_atomic_load(&b[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
template <int AG>
__device__ inline void phase_body(const double* src, double* dst, const uint32_t* idx, unsigned M, int K, unsigned gid, unsigned nth) {
  for (unsigned t = gid; t < M; t += nth) {
    double acc = 0.0;
    for (int k = 0; k < K; ++k) { const uint32_t j = idx[(size_t)k * M + t]; acc += AG ? ld_ag(&src[j]) : src[j]; }
    const double v = acc * (1.0 / K) + 1.0;
    if (AG) st_ag(&dst[t], v); else dst[t] = v;
  }
}
__global__ void __launch_bounds__(256) k_persist(double* a, double* b, const uint32_t* idx, unsigned M, int K, int phases, unsigned* barw) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  double* src = a; double* dst = b;
  for (int p = 0; p < phases; ++p) { phase_body<1>(src, dst, idx, M, K, gid, nth); bar(barw, gridDim.x); double* t = src; src = dst; dst = t; }
}
__global__ void __launch_bounds__(256) k_phase(const double* src, double* dst, const uint32_t* idx, unsigned M, int K) {
  phase_body<0>(src, dst, idx, M, K, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
__global__ void k_idx(uint32_t* idx, unsigned M, int K) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < M * (unsigned)K) idx[i] = mixh(i, 77u) % M; }
__global__ void k_empty_bar(int phases, unsigned* barw) { for (int p = 0; p < phases; ++p) bar(barw, gridDim.x); }
int main() {
  const unsigned M = 200000; const int K = 10; const int phases = 2000;
  double *a, *b, *ra, *rb; uint32_t* idx; unsigned* barw;
  hipMalloc(&a, M * 8); hipMalloc(&b, M * 8); hipMalloc(&ra, M * 8); hipMalloc(&rb, M * 8); hipMalloc(&idx, (size_t)M * K * 4); hipMalloc(&barw, 256);
  k_idx<<<(M * K + 255) / 256, 256>>>(idx, M, K);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  // reference: one launch per phase
  hipMemset(ra, 0, M * 8); hipMemset(rb, 0, M * 8);
  hipEventRecord(e0); for (int p = 0; p < phases; ++p) k_phase<<<(M + 255) / 256, 256>>>(p & 1 ? rb : ra, p & 1 ? ra : rb, idx, M, K); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); printf("one launch per phase (%u blocks): %.2f us per phase\n", (M + 255) / 256, ms * 1e3 / phases);
  std::vector<double> ref(M), got(M); hipMemcpy(ref.data(), (phases & 1) ? rb : ra, M * 8, hipMemcpyDeviceToHost);
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persist, 256, 0); printf("k_persist: %d resident blocks of 256 per CU\n", occ);
  for (unsigned blocks : {128u, 256u, 512u, 768u, 1024u}) {
    if (blocks > 256u * (unsigned)occ) continue;
    hipMemset(a, 0, M * 8); hipMemset(b, 0, M * 8); hipMemset(barw, 0, 256);
    hipEventRecord(e0); k_persist<<<blocks, 256>>>(a, b, idx, M, K, phases, barw); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(got.data(), (phases & 1) ? b : a, M * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; for (unsigned i = 0; i < M; ++i) if (memcmp(&ref[i], &got[i], 8)) ++bad;
    printf("persistent, fence-free, %4u blocks: %.2f us per phase, values differing from the per-launch run: %zu of %u\n", blocks, ms * 1e3 / phases, bad, M);
    hipMemset(barw, 0, 256);
    hipEventRecord(e0); k_empty_bar<<<blocks, 256>>>(phases, barw); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("            barrier alone, %4u blocks: %.2f us\n", blocks, ms * 1e3 / phases);
  }
  return 0;
}
