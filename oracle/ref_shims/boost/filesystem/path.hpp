// Stand-in for Boost.Filesystem v2-style API (oracle/_ref only): the reference only stores, prints and splits paths.
#pragma once
#include <fstream>
#include <ostream>
#include <string>
#include <sys/stat.h>
namespace boost { namespace filesystem {
class path {
  std::string s;
 public:
  path() {}
  path(const char *p) : s(p) {}
  path(const std::string &p) : s(p) {}
  const char *c_str() const { return s.c_str(); }
  const std::string &string() const { return s; }
  const std::string &native() const { return s; }
  std::string file_string() const { return s; }
  bool empty() const { return s.empty(); }
  path filename() const {
    const std::size_t k = s.find_last_of('/');
    return path(k == std::string::npos ? s : s.substr(k + 1));
  }
  path leaf() const { return filename(); }
  path extension() const {
    const std::string f = filename().s;
    const std::size_t k = f.find_last_of('.');
    return path(k == std::string::npos || k == 0 ? std::string() : f.substr(k));
  }
  path stem() const {
    const std::string f = filename().s;
    const std::size_t k = f.find_last_of('.');
    return path(k == std::string::npos || k == 0 ? f : f.substr(0, k));
  }
  path operator/(const path &o) const { return path(s + "/" + o.s); }
  bool operator==(const path &o) const { return s == o.s; }
  friend std::ostream &operator<<(std::ostream &os, const path &p) { return os << p.s; }
};
inline bool exists(const path &p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
class ifstream : public std::ifstream {
 public:
  ifstream() {}
  explicit ifstream(const path &p, std::ios_base::openmode m = std::ios_base::in) : std::ifstream(p.c_str(), m) {}
  void open(const path &p, std::ios_base::openmode m = std::ios_base::in) { std::ifstream::open(p.c_str(), m); }
};
class ofstream : public std::ofstream {
 public:
  ofstream() {}
  explicit ofstream(const path &p, std::ios_base::openmode m = std::ios_base::out) : std::ofstream(p.c_str(), m) {}
  void open(const path &p, std::ios_base::openmode m = std::ios_base::out) { std::ofstream::open(p.c_str(), m); }
};
}}
