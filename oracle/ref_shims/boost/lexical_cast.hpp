// Stand-in for boost::lexical_cast (oracle/_ref only): stream conversion that must consume the whole input.
#pragma once
#include <boost/static_assert.hpp>
#include <sstream>
#include <stdexcept>
#include <string>
#include <typeinfo>
namespace boost {
class bad_lexical_cast : public std::bad_cast {
 public:
  const char *what() const noexcept override { return "bad lexical cast"; }
};
template <class Target, class Source> Target lexical_cast(const Source &s) {
  std::stringstream ss;
  ss.precision(17);
  Target t;
  if (!(ss << s) || !(ss >> t)) throw bad_lexical_cast();
  ss >> std::ws;
  if (!ss.eof()) throw bad_lexical_cast();
  return t;
}
template <> inline std::string lexical_cast<std::string, std::string>(const std::string &s) { return s; }
}
