// oracle/_stub/vbem — TEST INFRASTRUCTURE: boost::filesystem::path as far as CollapsedGibbsSampler.cpp uses it (a name that is joined and printed)
#pragma once
#include <string>
namespace boost { namespace filesystem { class path { std::string s_; public: path() {} path(const std::string& s) : s_(s) {} path(const char* s) : s_(s) {} const std::string& string() const { return s_; }
  path operator/(const path& o) const { return path(s_ + "/" + o.s_); } path operator/(const char* o) const { return path(s_ + "/" + o); } };
  inline bool exists(const path&) { return true; } inline bool create_directories(const path&) { return true; } inline bool is_directory(const path&) { return true; } } }
