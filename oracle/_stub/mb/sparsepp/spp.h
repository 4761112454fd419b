// oracle/_stub/mb — TEST INFRASTRUCTURE.  EquivalenceClassBuilder.hpp names spp::sparse_hash_map (sparsepp, absent) in its single-cell value type, which
// the bulk path never instantiates.
#pragma once
#include <unordered_map>
namespace spp { template <class K, class V> using sparse_hash_map = std::unordered_map<K, V>; }
