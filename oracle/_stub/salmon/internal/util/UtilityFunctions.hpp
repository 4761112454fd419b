// oracle/_stub/.../UtilityFunctions.hpp — TEST INFRASTRUCTURE.  The reference's header of this name pulls in SalmonUtils.hpp (Boost, spdlog, TBB,
// pufferfish); SBModel needs one thing from it, an integer power usable in constant expressions.
#pragma once
#include <cstdint>
#include <array>
#include <atomic>
#include <string>
#include <vector>
#include <iostream>
constexpr int64_t constExprPow(int64_t base, unsigned int e) { int64_t r = 1; while (e) { if (e & 1u) r *= base; base *= base; e >>= 1; } return r; }
