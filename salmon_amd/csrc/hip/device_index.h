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
  const uint32_t* gcpre = nullptr;      // G/C prefix of refseq per 32-nt word (fragment-GC bias: sq_gc_before)
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

// label-major CSR already resident in HBM (a ctx's staged eq-class export): off[E+1], tid[L], w[L], count[E]
struct sq_eq_dev_csr { uint64_t E, L; const uint64_t* off; const uint32_t* tid; const double* w; const unsigned long long* cnt; };
// EM over a host table (eq) or over a CSR that already lives on `device` (dv); em.hip
// arena_slot: where the caller keeps its persistent EM workspace (a ctx), or nullptr for a private one
// lent_stream: an idle hipStream_t of the caller to run on (nullptr: the session creates its own)
int sq_em_optimize_impl(int device, const sq_eq_table* eq, const sq_eq_dev_csr* dv, const sq_txp_in* txp, const sq_em_opts* o,
    double* alpha_out, sq_em_report* rep,
    void** arena_slot, void* lent_stream);
int sq_em_arena_reserve(void** slot, size_t bytes, size_t pinned_bytes, size_t pinned_plan_bytes);
void sq_em_arena_free(void* slot);
size_t sq_em_workspace_bytes(uint64_t E, uint64_t L, uint64_t M);
