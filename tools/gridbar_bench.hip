// tools/gridbar_bench.hip — cost of a hand-rolled grid barrier on MI355X (8 XCDs, L2s not coherent with each other):
// is a persistent multi-phase kernel (EM iteration, online mini-batch chain) cheaper than one launch per phase?
// Each phase writes one value per thread that another block reads in the next phase (so the barrier must really
// publish data across XCDs), then all blocks meet at a sense-reversing counter barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline void grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();                                                   // release: this block's stores -> device
    if (atomicAdd(count, 1u) == nblocks - 1) { atomicExch(count, 0u); __threadfence(); __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    else { while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1); }
    __threadfence();                                                   // acquire
  }
  __syncthreads();
}
__global__ void k_persist(double* a, double* b, unsigned n, int phases, unsigned* count, unsigned* gen, unsigned long long* bad) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  double* src = a; double* dst = b;
  for (int p = 0; p < phases; ++p) {
    for (unsigned i = gid; i < n; i += nth) { const unsigned j = (i + 7919u * 64u) % n; const double v = __builtin_nontemporal_load(&src[j]); dst[i] = v + 1.0; }   // reads another block's (other XCD's) output
    grid_barrier(count, gen, gridDim.x);
    double* t = src; src = dst; dst = t;
  }
  if (gid < n && src[gid] != (double)phases) atomicAdd(bad, 1ULL);
}
__global__ void k_phase(const double* src, double* dst, unsigned n) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (unsigned i = gid; i < n; i += nth) { const unsigned j = (i + 7919u * 64u) % n; dst[i] = src[j] + 1.0; }
}
int main() {
  const unsigned n = 1u << 18; const int phases = 2000;
  double *a, *b; unsigned *cnt, *gen; unsigned long long* bad;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&cnt, 4); hipMalloc(&gen, 4); hipMalloc(&bad, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  for (unsigned blocks : {64u, 256u, 512u}) {
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8); hipMemset(cnt, 0, 4); hipMemset(gen, 0, 4); hipMemset(bad, 0, 8);
    hipEventRecord(e0); k_persist<<<blocks, 256>>>(a, b, n, phases, cnt, gen, bad); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); unsigned long long hb = 0; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    printf("persistent, %3u blocks: %.2f us per phase (compute + grid barrier), wrong values: %llu\n", blocks, ms * 1e3 / phases, hb);
  }
  hipMemset(a, 0, n * 8);
  hipEventRecord(e0); for (int p = 0; p < phases; ++p) { k_phase<<<256, 256>>>(p & 1 ? b : a, p & 1 ? a : b, n); } hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("one launch per phase, 256 blocks: %.2f us per phase\n", ms * 1e3 / phases);
  return 0;
}
