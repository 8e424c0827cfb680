// Stand-in for Boost.Iostreams (oracle/_ref only): no compression filters; the stream forwards to what is pushed.
#pragma once
#include <iostream>
#include <memory>
namespace boost { namespace iostreams {
struct input {};
struct output {};
struct gzip_decompressor {};
struct gzip_compressor {};
template <class C> struct basic_null_sink {};
typedef basic_null_sink<char> null_sink;
template <class Mode> class filtering_stream : public std::iostream {
 public:
  filtering_stream() : std::iostream(nullptr) {}
  void push(const gzip_decompressor &) {}
  void push(const gzip_compressor &) {}
  void push(std::istream &s) { rdbuf(s.rdbuf()); }
  void push(std::ostream &s) { rdbuf(s.rdbuf()); }
  void reset() { rdbuf(nullptr); }
  void pop() { rdbuf(nullptr); }
  bool empty() const { return rdbuf() == nullptr; }
  void clear_chain() { rdbuf(nullptr); }
  using std::iostream::clear;
};
template <class Sink> class stream : public std::ostream {
 public:
  stream() : std::ostream(nullptr) {}
  explicit stream(const Sink &) : std::ostream(nullptr) {}
  void open(const Sink &) {}
};
}}
