// oracle/_stub/Kmer.hpp — TEST INFRASTRUCTURE.  Stand-in for pufferfish's combinelib::kmers::Kmer (include/Kmer.hpp of the absent pufferfish tree),
// just enough for the reference's src/model/SBModel.cpp to compile into oracle/_ref: a k-mer of up to 32 bases in one 64-bit word, the FIRST
// character in the highest two bits of the k-mer (what the jellyfish mer_dna this class replaced does, and what SBModel's shifts
// 2 * contextLength - 2 * (i + 1) assume: position 0 of the context is the top pair), A = 0, C = 1, G = 2, T = 3.
#pragma once
#include <cstdint>
namespace combinelib { namespace kmers {
template <uint64_t K, uint64_t CID>
class Kmer {
public:
  static void k(int kIn) { k_ = kIn; }   // the length is a property of the class (one per class id CID), as in pufferfish: SBModel sets it once, every SBMer has it
  bool fromChars(const char* s) {
    w_ = 0; bool ok = true;
    for (int i = 0; i < k_; ++i) { uint64_t c = 0; switch (s[i]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; default: ok = false; }
      w_ = (w_ << 2) | c; }
    return ok;
  }
  void rc() { uint64_t r = 0, v = w_; for (int i = 0; i < k_; ++i) { r = (r << 2) | (3u - (v & 3u)); v >>= 2; } w_ = r; }
  uint64_t get_bits(int shift, int width) const { return (w_ >> shift) & ((1ull << width) - 1ull); }
  uint64_t word() const { return w_; }
private:
  uint64_t w_ = 0; static inline int k_ = (int)K;
};
}}
