// oracle/_stub/SpinLock.hpp — TEST INFRASTRUCTURE.  FragmentLengthDistribution.hpp includes RapMap's spin lock (pufferfish / RapMap are absent from
// the reference tree); cacheCMF only needs try_lock / unlock.
#pragma once
#include <atomic>
class SpinLock {
  std::atomic_flag f_ = ATOMIC_FLAG_INIT;
public:
  void lock() { while (f_.test_and_set(std::memory_order_acquire)) {} }
  bool try_lock() { return !f_.test_and_set(std::memory_order_acquire); }
  void unlock() { f_.clear(std::memory_order_release); }
};
