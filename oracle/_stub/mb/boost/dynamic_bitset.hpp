// oracle/_stub/poly — TEST INFRASTRUCTURE.  Stand-ins on the include path of the polytope pin only (oracle/Makefile, ref_polytope_shim.cpp): they let
// /root/reference/include/salmon/internal/quant/{TranscriptCluster,ClusterForest}.hpp compile where they lie, without Boost.
// boost::dynamic_bitset<>: a resizable bit vector; the header uses the size constructor, operator[] and assignment.
#pragma once
#include <cstddef>
#include <vector>
namespace boost { template <class B = unsigned long> class dynamic_bitset { std::vector<bool> v_; public: dynamic_bitset() {} explicit dynamic_bitset(size_t n) : v_(n, false) {}
  std::vector<bool>::reference operator[](size_t i) { return v_[i]; } bool operator[](size_t i) const { return v_[i]; } size_t size() const { return v_.size(); } }; }
