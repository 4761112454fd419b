// oracle/ref_alnmodel_shim.cpp — TEST INFRASTRUCTURE.  C entry points around the reference's CIGAR-based alignment error model, compiled from where the
// sources lie under /root/reference (never copied) into oracle/_ref/libalnmodel_ref.so by oracle/Makefile:
//   src/alignment/AlignmentModel.cpp    AlignmentModel::logLikelihood / update (ReadPair and UnpairedRead forms, and the record walks behind them)
//   src/alignment/AlignmentCommon.cpp   setBasesFromCIGAROp_
//   include/salmon/internal/util/AtomicMatrix.hpp (header), SalmonStringUtils.hpp (samToTwoBit)
// htslib's bam1_t, spdlog, TBB's concurrent_vector, SalmonUtils.hpp (incLoopLog), Transcript / ReadPair / UnpairedRead are stood in for by oracle/_stub/aln.
// Pins the checker's err_like_rec / err_update_rec and the per-mini-batch application of the increments (oracle.cpp) — tests/test_alnmodel_pin.py.
#include "salmon/internal/alignment/AlignmentModel.hpp"
#include "salmon/internal/alignment/ReadPair.hpp"
#include "salmon/internal/alignment/UnpairedRead.hpp"
#include "salmon/internal/model/Transcript.hpp"
#include <cstdint>
#include <cstring>
#include <vector>
namespace {
// a record as htslib lays it out: name, CIGAR, bases two per byte, qualities
bam1_t* make_rec(int32_t pos, uint16_t flag, const uint32_t* cig, uint32_t ncig, const uint8_t* seq, int32_t len) {
  bam1_t* b = bam_init1(); b->core.pos = pos; b->core.flag = flag; b->core.n_cigar = ncig; b->core.l_qseq = len; b->core.l_qname = 4;
  const size_t bytes = 4 + (size_t)ncig * 4 + (size_t)(len + 1) / 2 + (size_t)len + 8; b->data = (uint8_t*)calloc(bytes, 1); memcpy(b->data, "r\0\0\0", 4);
  memcpy(b->data + 4, cig, (size_t)ncig * 4); uint8_t* s = b->data + 4 + (size_t)ncig * 4; static const uint8_t code[4] = {1, 2, 4, 8};
  for (int32_t i = 0; i < len; ++i) s[i >> 1] |= (uint8_t)(code[seq[i] & 3] << ((~i & 1) << 2));
  return b;
}
struct Model { AlignmentModel m; Transcript t; Model(uint32_t bins) : m(1.0, bins) {} };
}
extern "C" {
void* ref_aln_model_new(uint32_t bins, const uint8_t* txp_bases, uint32_t txp_len) {
  Model* M = new Model(bins); M->t.RefLength = txp_len; M->t.RefName = "t"; M->t.SAMSequence_.assign((txp_len + 1) / 2 + 1, 0); static const uint8_t code[4] = {1, 2, 4, 8};
  for (uint32_t i = 0; i < txp_len; ++i) M->t.SAMSequence_[i >> 1] |= (uint8_t)(code[txp_bases[i] & 3] << ((!(i & 1)) << 2));
  return M;
}
void ref_aln_model_free(void* h) { delete (Model*)h; }
// kind: 0 = proper pair (records in file order: read1 first), 1 = left orphan, 2 = right orphan, 3 = single-end.  Returns logLikelihood(aln) and, when do_update, applies update(aln, p, mass)
double ref_aln_model_eval(void* h, int kind, int32_t pos1, const uint32_t* cig1, uint32_t n1, const uint8_t* seq1, int32_t len1,
                          int32_t pos2, const uint32_t* cig2, uint32_t n2, const uint8_t* seq2, int32_t len2, int do_update, double p, double mass) {
  Model* M = (Model*)h; double ll = 0.0;
  bam1_t* a = make_rec(pos1, 0, cig1, n1, seq1, len1); bam1_t* b = kind == 0 ? make_rec(pos2, 0, cig2, n2, seq2, len2) : nullptr;
  if (kind == 3) { UnpairedRead u; u.read = a; ll = M->m.logLikelihood(u, u, M->t); if (do_update) M->m.update(u, u, M->t, p, mass); }
  else { ReadPair r; r.read1 = a; r.read2 = b; r.orphanStatus = kind == 0 ? salmon::utils::OrphanStatus::Paired : (kind == 1 ? salmon::utils::OrphanStatus::LeftOrphan : salmon::utils::OrphanStatus::RightOrphan);
    ll = M->m.logLikelihood(r, r, M->t); if (do_update) M->m.update(r, r, M->t, p, mass); }
  bam_destroy1(a); if (b) bam_destroy1(b);
  return ll;
}
}
