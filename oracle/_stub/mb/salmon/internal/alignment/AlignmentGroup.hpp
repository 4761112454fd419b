// oracle/_stub/mb/.../AlignmentGroup.hpp — TEST INFRASTRUCTURE.  The reference's AlignmentGroup.hpp pulls in htslib and cereal; processMiniBatch uses a read's
// group as a vector of alignments (alignments(), size()) — include/salmon/internal/alignment/AlignmentGroup.hpp:30-60.
#pragma once
#include <vector>
template <typename FragT> class AlignmentGroup { public: std::vector<FragT> alns; std::vector<FragT>& alignments() { return alns; } size_t size() const { return alns.size(); } bool& isUniquelyMapped() { return uniq_; } private: bool uniq_ = true; };
