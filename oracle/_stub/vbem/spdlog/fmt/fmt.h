// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// spdlog's logger: messages are dropped (the iteration count of the optimiser's closing message is kept).
#include <cstddef>
#pragma once
#include <memory>
#include <string>
namespace spdlog { class logger { public: size_t last_iter = 0; double last_rel = 0.0;
  void info(const char*, size_t it, double rel) { last_iter = it; last_rel = rel; }   // "iteration = {} | max rel diff. = {}": the optimiser's last word is its iteration count
  template <class... A> void info(const A&...) {} template <class... A> void warn(const A&...) {} template <class... A> void error(const A&...) {}
  template <class... A> void critical(const A&...) {} void flush() {} }; inline void drop_all() {} }
