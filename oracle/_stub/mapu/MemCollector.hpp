#pragma once
#include "Util.hpp"
template <class IndexT> struct MemCollector { void configureMemClusterer(uint32_t) {} void setConsensusFraction(double) {} void setHitFilterPolicy(pufferfish::util::HitFilterPolicy) {} void setAltSkip(uint32_t) {} void setChainSubOptThresh(double) {} };
