// hip/device_index.h — HBM-resident mirror of sq_index (query structures only).
#pragma once
#include <hip/hip_runtime.h>
#include "../host/index.h"

struct sq_device_index {
  int device = -1;
  uint64_t bytes = 0;
  sq_dict_view dict;            // device pointers
  uint32_t k = 0, first_decoy = 0, num_refs = 0;
  const uint64_t* ref_accum = nullptr;  // [nrefs+1]
  const uint32_t* ref_len = nullptr;    // [nrefs]
  const uint32_t* ref_clen = nullptr;
  const uint64_t* refseq = nullptr;
  const uint64_t* ctab_off = nullptr;
  const uint64_t* ctab = nullptr;
  std::vector<void*> allocs;
};

#define SQ_HIP_CHECK(expr)                                                                     \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      sq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return SQ_ERR_DEVICE;                                                                    \
    }                                                                                          \
  } while (0)
