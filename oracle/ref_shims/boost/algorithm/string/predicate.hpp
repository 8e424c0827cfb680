#pragma once
#include <algorithm>
#include <cctype>
#include <string>
namespace boost {
inline bool starts_with(const std::string &s, const std::string &p) { return s.compare(0, p.size(), p) == 0; }
inline bool ends_with(const std::string &s, const std::string &p) { return s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0; }
inline bool iequals(const std::string &a, const std::string &b) {
  return a.size() == b.size() && std::equal(a.begin(), a.end(), b.begin(), [](char x, char y) { return std::tolower((unsigned char)x) == std::tolower((unsigned char)y); });
}
namespace algorithm { using boost::starts_with; using boost::ends_with; using boost::iequals; }
}
