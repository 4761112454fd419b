#pragma once
#include <cstdint>
#define KSW_EZ_RIGHT 0x08
#define KSW_EZ_SCORE_ONLY 0x01
namespace ksw2pp { struct KSW2Config { int dropoff, gapo, gape, bandwidth, flag; }; struct KSW2Aligner { KSW2Aligner(int8_t = 2, int8_t = 4) {} KSW2Config c; KSW2Config& config() { return c; } }; }
