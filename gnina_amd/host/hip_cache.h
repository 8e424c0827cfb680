// hip_cache.h -- C++ host adapters for the Vina / Monte-Carlo seam of gnina, above the C ABI.
//
//   HipCache         <->  cache : igrid              gninasrc/lib/cache.h:38-62, igrid.h:32-46
//   HipQuasiNewton   <->  quasi_newton               gninasrc/lib/quasi_newton.h:29-40, quasi_newton.cpp:49-83
//
// gnina already treats "BFGS given (model, igrid, conf)" as an offload unit: quasi_newton::operator()
// dynamic_casts the igrid to cache_gpu / non_cache_gpu and then runs the whole minimisation on the device
// (quasi_newton.cpp:52-72).  HipCache is that kind of igrid for the MI355X engine:
//   * through the plain igrid interface (eval / eval_deriv on m.coords) it is a drop-in for `cache` -- gnina's own
//     CPU bfgs<> and monte_carlo can run on it unchanged (tests/cpp/test_igrid_dropin.cpp does exactly that with the
//     reference's code);
//   * HipQuasiNewton recognises it and hands the whole minimisation to mi_vina_bfgs_batch.
// Inside a gnina build define MI_GNINA_WITH_GNINA_HEADERS (real model / igrid / conf); stand-alone the stand-ins of
// gnina_types.h are used and conformations are flat [7 + T] vectors.
#pragma once
#include <memory>
#include <vector>

#include "../../include/mi_gnina.h"
#ifdef MI_GNINA_WITH_GNINA_HEADERS
#include "cache.h"
#include "igrid.h"
#include "model.h"
#include "quasi_newton.h"
#else
#include "gnina_types.h"
#endif

namespace gnina_amd {

// Flat form of a ligand's torsion tree, pair list and local coordinates: owns the arrays an mi_ligand_desc points to.
struct LigandArrays {
  std::vector<int32_t> smt, parent, abeg, aend, pairs;
  std::vector<float> local_xyz, rel_origin, rel_axis;
  mi_ligand_desc desc() const;
  int n_torsions() const { return (int)parent.size() - 1; }
};
#ifdef MI_GNINA_WITH_GNINA_HEADERS
// model.ligands[0] (heterotree<rigid_body> of tree<segment>, tree.h:313-401) in model order -> LigandArrays.
// Ligand atoms must start at atom 0 of the model (no flexible residues in front).
LigandArrays ligand_arrays_from_model(const model &m);
// conf <-> flat [position 3][orientation a b c d][torsions]; change -> [force 3][torque 3][torsions]
std::vector<float> flatten(const conf &c);
void unflatten(const std::vector<float> &x, conf &c);
void unflatten(const std::vector<float> &g, change &c);
#endif

class HipCache : public igrid {
  std::shared_ptr<mi_vina> v_;
  mutable std::vector<int32_t> smt_;  // smina types of the movable atoms of the model being evaluated (refreshed per call)
  void types_of(const model &m) const;
  mutable std::vector<float> xyz_, forces_;
  fl slope_ = 1e3;
  bool have_ligand_ = false, have_user_grid_ = false;

 public:
  // cache(scoring_function_version, gd, slope) + populate(m, p, atom_types_needed, ...) (cache.cpp:44-48,104-184):
  // default Vina terms / weights (main.cpp:1324-1329), receptor = m.grid_atoms, one grid per needed ligand type.
  // user_grid_text: content of the --user_grid file (main.cpp:1342-1350) or null; ug_scaling_factor as computed there.
  HipCache(const model &m, const grid_dims &gd, const std::vector<smt> &atom_types_needed, fl slope = 1e3,
           const std::string *user_grid_text = nullptr, fl ug_scaling_factor = 1);
  bool has_user_grid() const { return have_user_grid_; }

  fl eval(model &m, fl v) const override;
  fl eval_deriv(model &m, fl v, const grid &user_grid) const override;
  // (skip_interacting_pairs / adjust_center / get_center / move_receptor: igrid's defaults, like `cache`)

  // the offload unit: the ligand's tree goes to the device once, then whole minimisations / searches run there
  void set_ligand(const LigandArrays &lig);
  bool has_ligand() const { return have_ligand_; }
  // quasi_newton::operator() on the device for one conformation (flat [7 + T], in place); returns the energy
  fl bfgs(std::vector<float> &conf, std::vector<float> &grad, const vec &v, unsigned maxiters) const;
  mi_vina *handle() const { return v_.get(); }
};

// quasi_newton (quasi_newton.cpp:49-83): the device BFGS when the igrid is a HipCache with a ligand, else nothing to
// do here -- inside gnina the `else` branch is the unchanged CPU code.
class HipQuasiNewton {
  minimization_params params;

 public:
  explicit HipQuasiNewton(const minimization_params &p) : params(p) {}
  // returns false when `ig` is not a HipCache (the caller falls back to gnina's own quasi_newton)
  bool operator()(igrid &ig, std::vector<float> &conf, std::vector<float> &grad, const vec &v, fl &energy) const;
#ifdef MI_GNINA_WITH_GNINA_HEADERS
  // the reference's signature (quasi_newton.h:38-39): out.c / out.e / g are updated like quasi_newton::operator()
  bool operator()(model &m, const precalculate &p, igrid &ig, output_type &out, change &g, const vec &v,
                  const grid &user_grid) const;
#endif
};

}  // namespace gnina_amd
