// oracle/ref_edlib_shim.cpp — TEST INFRASTRUCTURE.  A C entry point around the reference's own infix aligner, so the
// tests can call the real thing: this file is compiled together with /root/reference/src/edlib.cpp (read where it
// lies, never copied) into oracle/_ref/libedlib_ref.so by oracle/Makefile.  It pins the a5 arithmetic (SURVEY.md
// §8a row a5): the restatement in oracle.cpp (infix_align) and the HIP kernel must return what this returns.
#include "salmon/vendor/edlib.h"
extern "C" int ref_edlib_infix(const char* q, int n, const char* t, int m, int k, int* ed, int* start, int* end, int* num_locations) {
  EdlibAlignResult r = edlibAlign(q, n, t, m, edlibNewAlignConfig(k, EDLIB_MODE_HW, EDLIB_TASK_LOC));
  int ok = 0;
  if (r.editDistance >= 0 && r.numLocations > 0) { *ed = r.editDistance; *start = r.startLocations[0]; *end = r.endLocations[0]; *num_locations = r.numLocations; ok = 1; }
  edlibFreeAlignResult(r);
  return ok;
}
