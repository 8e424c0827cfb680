// Stand-in for Boost.Regex (oracle/_ref only): the term-name patterns of everything.h are plain ECMAScript.
#pragma once
#include <regex>
#include <string>
namespace boost {
class regex : public std::regex {
 public:
  enum flag_type_ { perl = 0 };
  regex() {}
  regex(const char *p, flag_type_ = perl) : std::regex(p) {}
  regex(const std::string &p, flag_type_ = perl) : std::regex(p) {}
  regex &assign(const char *p, flag_type_ = perl) { std::regex::assign(p); return *this; }
  regex &assign(const std::string &p, flag_type_ = perl) { std::regex::assign(p); return *this; }
};
typedef std::smatch smatch;
using std::regex_match;
using std::regex_search;
}
