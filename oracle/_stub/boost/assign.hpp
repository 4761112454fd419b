// oracle/_stub/boost/assign.hpp — TEST INFRASTRUCTURE (Boost is absent; FragmentLengthDistribution.cpp includes this header and uses nothing of it).
#pragma once
