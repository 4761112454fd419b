// hip/inflate_dev.hip — [r5] BGZF members inflated on the device: a wave per member (inflate_core.h says how a wave decodes), the member's CRC-32 checked.  Used by the device FASTQ reader (fastq_dev.hip) for BGZF files:
// the compressed bytes cross PCIe (a third of the text), the text is born in HBM where the record splitter reads it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../host/index.h"
#include "inflate_core.h"
#include "inflate_dev.h"

namespace {
constexpr int INF_WAVES = 4;   // waves (members) per block
__global__ void __launch_bounds__(64 * INF_WAVES) k_bgzf_inflate(const uint8_t* __restrict__ comp, const sq_bgzf_member* __restrict__ mem, uint32_t nmem, uint8_t* __restrict__ text,
                                                                    uint32_t* __restrict__ status /* both start as 0xFFFFFFFF; [0] = index + 1 of the first bad member (atomicMin), [1] = (index + 1) << 4 | what was wrong with it (atomicMin) */) {
  __shared__ sqinf::Tables s_tab[INF_WAVES]; __shared__ uint32_t s_crc[256];
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_crc[i] = sqinf::crc32_entry(i);
  __syncthreads();
  // the wave's index as a uniform value: everything the decoder derives from it lives in scalar registers
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), m = blockIdx.x * INF_WAVES + wave;
  if (m >= nmem) return;
  const sq_bgzf_member M = mem[m];
  if (M.flags & SQ_BGZF_LINE_END) { if ((threadIdx.x & 63) < M.isize) text[M.voff + (threadIdx.x & 63)] = '\n'; return; }   // the line end a file without a last one gets (the reader's pseudo-member)
  int rc = M.csize ? sqinf::inflate_member(comp + M.coff, M.csize, text + M.voff, M.isize, s_tab[wave]) : (M.isize ? (int)sqinf::INF_EOF_INPUT : (int)sqinf::INF_OK);   // text without a stream: damaged
  if (rc == sqinf::INF_OK && sqinf::crc32_wave(s_crc, text + M.voff, M.isize) != M.crc) rc = 8;
  // [r6] index and cause leave in ONE word ((index + 1) << 4 | cause, atomicMin): the cause reported is the first bad member's whichever wave gets there first
  if (rc != sqinf::INF_OK && (threadIdx.x & 63) == 0) { (void)atomicMin(&status[0], m + 1); (void)atomicMin(&status[1], ((m + 1) << 4) | ((uint32_t)rc & 15u)); }
}
}  // namespace

const char* sq_bgzf_status_text(uint32_t what) {   // `what`: status[1] as the kernel left it
  switch (what & 15u) {
    case sqinf::INF_EOF_INPUT: return "truncated BGZF member";
    case 8: return "BGZF checksum mismatch";
    case sqinf::INF_OUTPUT_SIZE: return "corrupt BGZF member (its text is not the size its trailer names)";
    default: return "corrupt BGZF member";
  }
}
int sq_bgzf_inflate_launch(const uint8_t* d_comp, const sq_bgzf_member* d_mem, uint32_t nmem, uint8_t* d_text, uint32_t* d_status, hipStream_t st) {
  if (!nmem) return SQ_OK;
  k_bgzf_inflate<<<(nmem + INF_WAVES - 1) / INF_WAVES, 64 * INF_WAVES, 0, st>>>(d_comp, d_mem, nmem, d_text, d_status);
  return hipGetLastError() == hipSuccess ? SQ_OK : SQ_ERR_DEVICE;
}

// ---- test hooks -------------------------------------------------------------------------------------------------------------------------------
// the decoder's source on the host (the same header, compiled for the CPU): how tests check it against zlib where there is no GPU.  Not a path of the product.
extern "C" int sq_debug_inflate_core_host(const uint8_t* comp, uint64_t csize, uint8_t* out, uint32_t isize, uint32_t* crc_out) {
  static sqinf::Tables T; static uint32_t tab[256]; static bool have = false;
  if (!have) { for (uint32_t i = 0; i < 256; ++i) tab[i] = sqinf::crc32_entry(i); have = true; }
  const int rc = sqinf::inflate_member(comp, (size_t)csize, out, isize, T);
  if (rc == sqinf::INF_OK && crc_out) *crc_out = sqinf::crc32_wave(tab, out, isize);
  return rc;
}
// members (raw deflate streams back to back in `comp`, descriptors in `mem`) through the device kernel; text and status come back to the host
extern "C" int sq_debug_bgzf_inflate(int device, const uint8_t* comp, uint64_t comp_bytes, const void* members, uint32_t nmem, uint8_t* text, uint64_t text_bytes, uint32_t* status2) {
  const sq_bgzf_member* mem = (const sq_bgzf_member*)members;
  if (!comp || !mem || !text || !status2) { sq_set_error("sq_debug_bgzf_inflate: bad arguments"); return SQ_ERR_ARG; }
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); sq_set_error("no HIP device %d", device); return SQ_ERR_DEVICE; }
  void *dc = nullptr, *dm = nullptr, *dt = nullptr, *ds = nullptr; int rc = SQ_OK;
  if (hipMalloc(&dc, comp_bytes + 512) != hipSuccess || hipMalloc(&dm, (size_t)nmem * sizeof(sq_bgzf_member) + 16) != hipSuccess || hipMalloc(&dt, text_bytes + 64) != hipSuccess || hipMalloc(&ds, 16) != hipSuccess) rc = SQ_ERR_NOMEM;
  const uint32_t st0[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
  if (!rc && (hipMemcpy(dc, comp, comp_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dm, mem, (size_t)nmem * sizeof(sq_bgzf_member), hipMemcpyHostToDevice) != hipSuccess ||
              hipMemcpy(ds, st0, 8, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dt, 0, text_bytes) != hipSuccess)) rc = SQ_ERR_DEVICE;
  if (!rc) rc = sq_bgzf_inflate_launch((const uint8_t*)dc, (const sq_bgzf_member*)dm, nmem, (uint8_t*)dt, (uint32_t*)ds, nullptr);
  if (!rc && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(text, dt, text_bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(status2, ds, 8, hipMemcpyDeviceToHost) != hipSuccess)) rc = SQ_ERR_DEVICE;
  if (!rc) status2[1] = status2[0] == 0xFFFFFFFFu ? 0u : (status2[1] & 15u);   // the caller's view: [0] = index + 1 of the first bad member, [1] = its cause
  for (void* p : {dc, dm, dt, ds}) if (p) (void)hipFree(p);
  if (rc == SQ_ERR_DEVICE) sq_set_error("device failure in sq_debug_bgzf_inflate: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}
