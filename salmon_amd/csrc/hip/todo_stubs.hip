// hip/todo_stubs.hip — entry points declared in salmon_hip.h whose kernels are not written yet.
// They fail loudly (never silently fall back).
#include "device_index.h"
extern "C" int sq_bootstrap_dev(int, const sq_eq_table*, const sq_txp_in*, const sq_em_opts*, uint32_t, uint64_t, uint64_t, sq_replicate_cb, void*) {
  sq_set_error("sq_bootstrap_dev: not implemented yet (SURVEY.md §8a row a16)"); return SQ_ERR_STATE; }
extern "C" int sq_gibbs_dev(int, const sq_eq_table*, const sq_txp_in*, const sq_gibbs_opts*, const double*, uint32_t, uint64_t, uint64_t, sq_replicate_cb, void*) {
  sq_set_error("sq_gibbs_dev: not implemented yet (SURVEY.md §8a row a17)"); return SQ_ERR_STATE; }
