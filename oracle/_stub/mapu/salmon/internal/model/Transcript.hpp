// stand-in: the one member of Transcript that SalmonMappingUtils.hpp reads
#pragma once
#include <cstdint>
struct Transcript { bool decoy = false; bool isDecoy() const { return decoy; } uint32_t RefLength = 0; };
