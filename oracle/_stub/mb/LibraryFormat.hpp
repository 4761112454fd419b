// oracle/_stub/mb/LibraryFormat.hpp — TEST INFRASTRUCTURE.  Stand-in for pufferfish's <LibraryFormat.hpp> (COMBINE-lab/pufferfish @ ace68c1c, absent from
// /root/reference; the reference's include/salmon/internal/model/LibraryFormat.hpp only forwards to it).  The class as the reference's own
// src/model/LibraryFormat.cpp defines its members (that file is compiled as it lies) and as tests/LibraryTypeTests.cpp:11-36 uses it: three enums, the
// constructor, check(), formatID() / formatFromID() / maxLibTypeID().  The id layout (type | orientation << 1 | strandedness << 3) is this repository's
// (include/salmon_hip.h: sq_aln.format_id); the reference's tests only require that it round-trips.  On the include path of the mini-batch pin only.
#pragma once
#include <cstdint>
#include <iostream>
#include <string>
enum class ReadType : std::uint8_t { SINGLE_END = 0, PAIRED_END = 1 };
enum class ReadOrientation : std::uint8_t { SAME = 0, AWAY = 1, TOWARD = 2, NONE = 3 };
enum class ReadStrandedness : std::uint8_t { SA = 0, AS = 1, S = 2, A = 3, U = 4 };
class LibraryFormat {
public:
  LibraryFormat(ReadType type_in, ReadOrientation orientation_in, ReadStrandedness strandedness_in);
  ReadType type; ReadOrientation orientation; ReadStrandedness strandedness;
  bool check();
  friend std::ostream& operator<<(std::ostream& os, const LibraryFormat& lf);
  inline std::uint8_t formatID() const { return (std::uint8_t)((std::uint8_t)type | ((std::uint8_t)orientation << 1) | ((std::uint8_t)strandedness << 3)); }
  inline static LibraryFormat formatFromID(std::uint8_t id) { return LibraryFormat((ReadType)(id & 1), (ReadOrientation)((id >> 1) & 3), (ReadStrandedness)(id >> 3)); }
  inline static constexpr std::uint8_t maxLibTypeID() { return 39; }   // PAIRED_END | NONE << 1 | U << 3
  std::string toString() const { return ""; }
};
inline bool operator==(const LibraryFormat& a, const LibraryFormat& b) { return a.type == b.type && a.orientation == b.orientation && a.strandedness == b.strandedness; }
