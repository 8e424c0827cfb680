#include "typed_atoms.h"

#include <cstring>
#include <fstream>
#include <stdexcept>

#include "../../include/mi_gnina.h"

namespace gnina_amd {

namespace {
struct atom_info {  // gninatyper.cpp:30-36
  float x, y, z;
  int32_t type;
};
static_assert(sizeof(atom_info) == 16, "gninatypes record is 16 bytes");
}  // namespace

TypedAtoms read_gninatypes(const std::string &path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("could not open " + path);
  const std::streamoff bytes = f.tellg();
  if (bytes % (std::streamoff)sizeof(atom_info) != 0)
    throw std::runtime_error(path + ": size is not a multiple of the 16-byte gninatypes record");
  f.seekg(0);
  std::vector<atom_info> rec((size_t)bytes / sizeof(atom_info));
  if (!rec.empty()) f.read(reinterpret_cast<char *>(rec.data()), bytes);
  if (!f) throw std::runtime_error("short read on " + path);
  TypedAtoms out;
  out.xyz.reserve(rec.size() * 3);
  out.smt.reserve(rec.size());
  for (size_t i = 0; i < rec.size(); i++) {
    if (rec[i].type < 0 || rec[i].type >= 28)
      throw std::runtime_error(path + ": atom " + std::to_string(i) + " has smina type " + std::to_string(rec[i].type) +
                               " outside 0..27");
    out.xyz.push_back(rec[i].x);
    out.xyz.push_back(rec[i].y);
    out.xyz.push_back(rec[i].z);
    out.smt.push_back(rec[i].type);
  }
  return out;
}

void write_gninatypes(const std::string &path, const TypedAtoms &atoms) {
  if (atoms.xyz.size() != 3 * atoms.smt.size()) throw std::runtime_error("TypedAtoms: xyz / smt size mismatch");
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("could not open " + path + " for writing");
  for (size_t i = 0; i < atoms.smt.size(); i++) {
    atom_info a{atoms.xyz[3 * i], atoms.xyz[3 * i + 1], atoms.xyz[3 * i + 2], atoms.smt[i]};
    f.write(reinterpret_cast<const char *>(&a), sizeof a);
  }
  if (!f) throw std::runtime_error("write failed on " + path);
}

}  // namespace gnina_amd

// C entry points (declared in include/mi_gnina.h) so that non-C++ callers share the reader
namespace {
thread_local std::string g_io_error;
}

extern "C" {

mi_status mi_read_gninatypes(const char *path, float *xyz, int32_t *smt, int capacity, int *n_atoms) {
  try {
    if (!path || !n_atoms) throw std::runtime_error("NULL argument");
    gnina_amd::TypedAtoms a = gnina_amd::read_gninatypes(path);
    *n_atoms = (int)a.size();
    if (xyz && smt) {
      if ((int)a.size() > capacity) throw std::runtime_error("capacity too small for " + std::string(path));
      if (!a.smt.empty()) {
        std::memcpy(xyz, a.xyz.data(), a.xyz.size() * sizeof(float));
        std::memcpy(smt, a.smt.data(), a.smt.size() * sizeof(int32_t));
      }
    }
    return MI_OK;
  } catch (const std::exception &e) {
    g_io_error = e.what();
    return MI_ERR_INVALID;
  }
}

mi_status mi_write_gninatypes(const char *path, const float *xyz, const int32_t *smt, int n_atoms) {
  try {
    if (!path || n_atoms < 0 || (n_atoms > 0 && (!xyz || !smt))) throw std::runtime_error("bad argument");
    gnina_amd::TypedAtoms a;
    a.xyz.assign(xyz, xyz + 3 * (size_t)n_atoms);
    a.smt.assign(smt, smt + n_atoms);
    gnina_amd::write_gninatypes(path, a);
    return MI_OK;
  } catch (const std::exception &e) {
    g_io_error = e.what();
    return MI_ERR_INVALID;
  }
}

const char *mi_io_last_error(void) { return g_io_error.c_str(); }
}
