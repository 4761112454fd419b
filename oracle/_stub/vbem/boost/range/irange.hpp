// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
#pragma once
#include <cstddef>
namespace boost { template <class T> struct irange_t { T b, e; struct it { T v; T operator*() const { return v; } it& operator++() { ++v; return *this; } bool operator!=(const it& o) const { return v != o.v; } };
  it begin() const { return it{b}; } it end() const { return it{e}; } };
template <class T> irange_t<T> irange(T b, T e) { return irange_t<T>{b, e}; } }
#ifndef BOOST_LIKELY
#define BOOST_LIKELY(x) __builtin_expect(!!(x), 1)
#define BOOST_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif
