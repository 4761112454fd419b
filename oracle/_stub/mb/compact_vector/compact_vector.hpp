// oracle/_stub/mb — TEST INFRASTRUCTURE.  Transcript.hpp names compact::vector<uint64_t, 1> (pufferfish's bit vector, absent) for its reduced-memory GC table;
// the mini-batch pin never builds that table (no sequence is attached to the transcripts, reduceGCMemory_ stays false): the members Transcript.hpp names, nothing behind them.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
namespace compact { template <class T, unsigned B> class vector { std::vector<uint64_t> v_; public: vector() {} void resize(size_t n) { v_.assign(n, 0); } uint64_t& operator[](size_t i) { return v_[i]; } uint64_t operator[](size_t i) const { return v_[i]; }
  void clear_mem() {} size_t size() const { return v_.size(); } uint64_t* get() { return v_.data(); } }; }
