// oracle/_stub/poly — TEST INFRASTRUCTURE.  Stand-ins on the include path of the polytope pin only (oracle/Makefile, ref_polytope_shim.cpp): they let
// /root/reference/include/salmon/internal/quant/{TranscriptCluster,ClusterForest}.hpp compile where they lie, without Boost.
#pragma once
