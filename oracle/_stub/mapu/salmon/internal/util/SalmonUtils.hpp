// stand-in: the one enum of SalmonUtils.hpp that SalmonMappingUtils.hpp names (in templates that are never instantiated here)
#pragma once
#include <sstream>
#include <cstdint>
namespace salmon { namespace utils { enum class MappingType : uint8_t { UNMAPPED = 0, LEFT_ORPHAN = 1, RIGHT_ORPHAN = 2, BOTH_ORPHAN = 3, PAIRED_MAPPED = 4, SINGLE_MAPPED = 5, DECOY = 6 }; } }
