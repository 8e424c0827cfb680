// gnina_types.h -- the few gnina types the DLScorer seam touches.
//
// Inside a gnina build define MI_GNINA_WITH_GNINA_HEADERS: the real headers are used
// (gninasrc/lib/dl_scorer.h, model.h, user_opts.h) and HipCNNScorer derives from gnina's own
// DLScorer.  Stand-alone (this repository: gnina cannot be built here) the stand-ins
// below carry exactly the members DLScorer::setLigand / setReceptor / score read or write
// (gninasrc/lib/dl_scorer.cpp:36-217, cnn_torch_scorer.cpp:105-198, model.cu:236-259), with the
// same names, so the adapter source is identical in both builds.
#pragma once

#ifdef MI_GNINA_WITH_GNINA_HEADERS
#include "dl_scorer.h"
#else
#include <cmath>
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

typedef float fl;               // gninasrc/lib/common.h:47
typedef std::size_t sz;
typedef int smt;                // smina_atom_type::type, atom_constants.h:45-75 (0..27)

struct float3 { float x, y, z; };
typedef float3 gfloat3;

struct vec {                    // common.h:87-91
  fl data[3];
  vec() : data{0, 0, 0} {}
  vec(fl x, fl y, fl z) : data{x, y, z} {}
  fl operator[](sz i) const { return data[i]; }
  fl &operator[](sz i) { return data[i]; }
  fl x() const { return data[0]; }
  fl y() const { return data[1]; }
  fl z() const { return data[2]; }
};
typedef std::vector<vec> vecv;

struct atom {                   // atom.h / atom_base.h: the fields the scorer reads
  smt sm = 0;
  vec coords;                   // fixed (rigid receptor) atoms keep their own coordinates
  bool iscov = false;
  bool is_hydrogen() const { return sm == 0 || sm == 1; }
};
typedef std::vector<atom> atomv;

struct grid_dim { fl begin = 0, end = 0; sz n = 0; };
struct grid_dims { grid_dim d[3]; grid_dim &operator[](sz i) { return d[i]; } };

struct ligand_node { sz begin = 0, end = 0; };
struct ligand { ligand_node node; };

// model.h: movable atoms = [flex residues ..., ligand ...], then inflex atoms; fixed atoms separate
struct model {
  atomv atoms;                  // movable + inflex, index-aligned with coords
  vecv coords;
  atomv grid_atoms;             // fixed receptor atoms
  sz m_num_movable_atoms = 0;
  std::vector<ligand> ligands;
  vecv minus_forces;
  const atomv &get_movable_atoms() const { return atoms; }
  const atomv &get_fixed_atoms() const { return grid_atoms; }
  const vecv &coordinates() const { return coords; }
  void clear_minus_forces() { minus_forces.assign(m_num_movable_atoms, vec()); }   // model.cu:236-246
  void add_minus_forces(const std::vector<gfloat3> &f) {                           // model.cu:247-259
    sz j = 0;
    for (sz i = 0; i < m_num_movable_atoms; i++)
      if (!atoms[i].is_hydrogen()) {
        minus_forces[i][0] += f[j].x; minus_forces[i][1] += f[j].y; minus_forces[i][2] += f[j].z;
        j++;
      }
  }
  void scale_minus_forces(fl s) { for (auto &v : minus_forces) { v[0] *= s; v[1] *= s; v[2] *= s; } }
  std::vector<vec> get_heavy_atom_movable_coords() const {
    std::vector<vec> out;
    for (sz i = 0; i < m_num_movable_atoms; i++)
      if (!atoms[i].is_hydrogen()) out.push_back(coords[i]);
    return out;
  }
};

enum cnn_scoring_level { CNNnone, CNNrescore, CNNrefinement, CNNmetrorescore, CNNmetrorefine, CNNall };

struct cnn_options {            // user_opts.h:36-64 (fields the scorer uses)
  std::vector<std::string> cnn_models;        // external model files
  std::vector<std::string> cnn_model_names;   // built-in model names
  vec cnn_center = vec(NAN, NAN, NAN);
  unsigned cnn_rotations = 0;
  cnn_scoring_level cnn_scoring = CNNrescore;
  bool verbose = false;
  unsigned seed = 0;
};

// igrid.h:32-46 (the Vina / Monte-Carlo seam) and the few types its users touch
struct grid {                   // grid.h: the optional user grid; never initialised here
  bool initialized() const { return false; }
};
struct igrid {
  virtual ~igrid() {}
  virtual fl eval(model &m, fl v) const = 0;                               // needs m.coords
  virtual fl eval_deriv(model &m, fl v, const grid &user_grid) const = 0;  // needs m.coords, sets m.minus_forces
  virtual bool skip_interacting_pairs() const { return false; }
  virtual void adjust_center(model &m) {}
  virtual vec get_center() const { return vec(0, 0, 0); }
  virtual bool move_receptor() { return false; }
};
struct minimization_params {    // common.h:50-61
  unsigned maxiters = 0;
};

struct usage_error : std::runtime_error { using std::runtime_error::runtime_error; };      // common.h
struct internal_error : std::runtime_error {   // common.h:270-274: (file, line); adapters put their message in `file`
  std::string file;
  unsigned line;
  internal_error(const std::string &file_, unsigned line_) : std::runtime_error(file_), file(file_), line(line_) {}
};

class DLScorer {                // dl_scorer.h:23-66, signatures verbatim
 protected:
  std::vector<float3> ligand_coords, receptor_coords;
  std::vector<smt> ligand_smtypes, receptor_smtypes;
  std::vector<int> ligand_map, receptor_map;
  std::size_t num_atoms = 0;
  vec current_center = vec(NAN, NAN, NAN);
  cnn_options cnnopts;
  virtual void setLigand(const model &m);
  virtual void setReceptor(const model &m);

 public:
  DLScorer() {}
  DLScorer(const cnn_options &opts) : cnnopts(opts) {}
  virtual ~DLScorer() {}
  virtual bool initialized() const = 0;
  virtual bool has_affinity() const = 0;
  virtual const cnn_options &options() const { return cnnopts; }
  virtual float score(model &m, float &variance) = 0;
  virtual float score(model &m, bool compute_gradient, float &affinity, float &loss, float &variance) = 0;
  virtual void set_center_from_model(model &m);
  virtual vec get_center() const { return current_center; }
  virtual void set_bounding_box(grid_dims &box) const = 0;
  virtual std::shared_ptr<DLScorer> fresh_copy() const = 0;
};
#endif  // MI_GNINA_WITH_GNINA_HEADERS
