// oracle/_stub/mb — TEST INFRASTRUCTURE: boost::filesystem as far as ReadLibrary.hpp and SalmonOpts.hpp name it (file names that the mini-batch pin never sets)
#pragma once
#include <string>
namespace boost { namespace filesystem { class path { std::string s_; public: path() {} path(const std::string& s) : s_(s) {} path(const char* s) : s_(s) {} const std::string& string() const { return s_; }
  path operator/(const path& o) const { return path(s_ + "/" + o.s_); } path operator/(const char* o) const { return path(s_ + "/" + o); } path extension() const { auto p = s_.rfind('.'); return path(p == std::string::npos ? "" : s_.substr(p)); } };
  inline bool exists(const path&) { return true; } inline bool create_directories(const path&) { return true; } inline bool is_directory(const path&) { return true; } inline bool is_regular_file(const path&) { return true; } inline bool is_empty(const path&) { return false; } } }
