// oracle/_stub/mb — TEST INFRASTRUCTURE.  Transcript.hpp names rank9b (pufferfish's rank structure, absent) for its reduced-memory GC table; never built here.
#pragma once
#include <cstdint>
class rank9b { public: rank9b(const uint64_t*, uint64_t) {} uint64_t rank(uint64_t) const { return 0; } };
