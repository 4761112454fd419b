// oracle/_stub/aln — TEST INFRASTRUCTURE.  Stand-ins on the include path of the alignment-model pin only (oracle/Makefile, ref_alnmodel_shim.cpp): they let
// /root/reference/src/alignment/AlignmentModel.cpp and AlignmentCommon.cpp compile where they lie, without htslib / spdlog / TBB / Boost / pufferfish.
#pragma once
#include <memory>
#include <iostream>
#include <sstream>
namespace spdlog { class logger { public: template <class... A> void warn(const A&...) {} template <class... A> void info(const A&...) {} template <class... A> void error(const A&...) {} }; }
