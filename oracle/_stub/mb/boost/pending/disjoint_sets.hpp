// oracle/_stub/poly — TEST INFRASTRUCTURE.  Stand-ins on the include path of the polytope pin only (oracle/Makefile, ref_polytope_shim.cpp): they let
// /root/reference/include/salmon/internal/quant/{TranscriptCluster,ClusterForest}.hpp compile where they lie, without Boost.
// boost::disjoint_sets<Rank, Parent>: union by rank with full path compression, as Boost publishes it (boost/pending/disjoint_sets.hpp +
// detail/disjoint_sets.hpp: make_set, find_set = find_representative_with_full_compression, link = link_sets on the two representatives).
#pragma once
#include <cstddef>
namespace boost { template <class RankPA, class ParentPA> class disjoint_sets { RankPA rank_; ParentPA parent_; public: disjoint_sets(RankPA r, ParentPA p) : rank_(r), parent_(p) {}
  template <class E> void make_set(E x) { parent_[x] = x; rank_[x] = 0; }
  template <class E> E find_set(E x) { E r = x; while (parent_[r] != (size_t)r) r = (E)parent_[r]; while (parent_[x] != (size_t)r) { E n = (E)parent_[x]; parent_[x] = r; x = n; } return r; }
  template <class E> void link(E x, E y) { x = find_set(x); y = find_set(y); if (x == y) return; if (rank_[x] > rank_[y]) parent_[y] = x; else { parent_[x] = y; if (rank_[x] == rank_[y]) ++rank_[y]; } } }; }
