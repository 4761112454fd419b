// stand-in for itlib::small_vector as SalmonMappingUtils.hpp uses it (a vector with an inline buffer it can fall back to)
#pragma once
#include <vector>
namespace itlib { template <class T, unsigned N = 32> struct small_vector : std::vector<T> { void revert_to_static() { this->shrink_to_fit(); } }; }
