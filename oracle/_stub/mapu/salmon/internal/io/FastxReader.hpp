#pragma once
#include <string>
namespace salmon { namespace io { namespace fastx { struct CompatReadSeq { std::string seq, name; }; struct CompatReadPair { CompatReadSeq first, second; }; } } }
