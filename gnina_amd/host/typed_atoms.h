// typed_atoms.h -- the typed-atom on-disk format of gnina's tool chain.
//
// `.gninatypes` (gninasrc/gninatyper/gninatyper.cpp:30-36,65-75): a headerless array of
//     struct atom_info { float x, y, z; int type; }        // 16 bytes, host endianness (little)
// where `type` is the smina atom type index (atom_constants.h:45-75, 0..27).  It is what libmolgrid's
// example providers read, and the cheapest way to hand real complexes to the scorer without OpenBabel:
// type once with gnina's `gninatyper`, score many times here.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace gnina_amd {

struct TypedAtoms {
  std::vector<float> xyz;     // [n][3]
  std::vector<int32_t> smt;   // [n]
  size_t size() const { return smt.size(); }
};

// Throws std::runtime_error on unreadable files, sizes that are not a multiple of 16 bytes, or type
// indices outside 0..27.
TypedAtoms read_gninatypes(const std::string &path);
void write_gninatypes(const std::string &path, const TypedAtoms &atoms);

}  // namespace gnina_amd
