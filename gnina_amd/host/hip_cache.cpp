#include "hip_cache.h"

#include <cstring>
#include <stdexcept>
#include <string>

namespace gnina_amd {

static void ok(mi_status st) {
  if (st != MI_OK) throw internal_error(mi_last_error(), 0);
}

mi_ligand_desc LigandArrays::desc() const {
  mi_ligand_desc d{};
  d.n_atoms = (int32_t)smt.size();
  d.smt = smt.data();
  d.local_xyz = local_xyz.data();
  d.n_nodes = (int32_t)parent.size();
  d.node_parent = parent.data();
  d.node_atom_begin = abeg.data();
  d.node_atom_end = aend.data();
  d.node_rel_origin = rel_origin.data();
  d.node_rel_axis = rel_axis.data();
  d.n_pairs = (int32_t)(pairs.size() / 2);
  d.pairs = pairs.data();
  return d;
}

HipCache::HipCache(const model &m, const grid_dims &gd_in, const std::vector<smt> &atom_types_needed, fl slope,
                   const std::string *user_grid_text, fl ug_scaling_factor)
    : slope_(slope) {
  mi_vina *v = mi_vina_create(nullptr, 8.0f, 32.0f);  // precalculate_linear(sf, 32) on the default terms
  if (!v) throw internal_error(mi_last_error(), 0);
  v_.reset(v, mi_vina_destroy);
  const atomv &fixed = m.get_fixed_atoms();
  std::vector<float> xyz(fixed.size() * 3);
  std::vector<int32_t> t(fixed.size());
  for (size_t i = 0; i < fixed.size(); i++) {
    for (int k = 0; k < 3; k++) xyz[3 * i + k] = fixed[i].coords[k];
    t[i] = (int32_t)fixed[i].sm;
  }
  ok(mi_vina_set_receptor(v, xyz.data(), t.data(), (int)t.size()));
  grid_dims gd = gd_in;
  float b[3], e[3];
  int32_t n[3];
  for (int i = 0; i < 3; i++) {
    b[i] = gd[i].begin;
    e[i] = gd[i].end;
    n[i] = (int32_t)gd[i].n;
  }
  if (user_grid_text) {  // --user_grid: the file main.cpp:1342-1350 reads; baked into the grids by populate
    float ub[3], ue[3];
    int32_t un[3];
    size_t cnt = 0;
    ok(mi_user_grid_parse(user_grid_text->data(), user_grid_text->size(), ub, ue, un, nullptr, 0, &cnt));
    std::vector<double> vals(cnt);
    ok(mi_user_grid_parse(user_grid_text->data(), user_grid_text->size(), ub, ue, un, vals.data(), cnt, &cnt));
    ok(mi_vina_set_user_grid(v, ub, ue, un, vals.data(), ug_scaling_factor));
    have_user_grid_ = true;
  }
  std::vector<int32_t> types(atom_types_needed.begin(), atom_types_needed.end());
  ok(mi_vina_build_cache(v, b, e, n, types.data(), (int)types.size(), slope));
  const atomv &mov = m.get_movable_atoms();
  smt_.resize(m.m_num_movable_atoms);
  for (size_t i = 0; i < smt_.size(); i++) smt_[i] = (int32_t)mov[i].sm;
}

// cache::eval reads the types of the model it is GIVEN (m.atoms[i].get(), cache.cpp:55-58), which need not be the model
// the grids were populated for -- as long as its atom types have grids.  The types are refreshed from `m` on every
// call (a few dozen ints), so reusing the igrid with another ligand gives that ligand's energies, like `cache`.
void HipCache::types_of(const model &m) const {
  const atomv &mov = m.get_movable_atoms();
  const size_t n = m.m_num_movable_atoms;
  if (m.coordinates().size() < n) throw internal_error("HipCache: the model has fewer coordinates than movable atoms", 0);
  smt_.resize(n);
  for (size_t i = 0; i < n; i++) smt_[i] = (int32_t)mov[i].sm;
}

// cache::eval (cache.cpp:50-63): sum over the movable heavy atoms of their type grid at m.coords
fl HipCache::eval(model &m, fl v) const {
  types_of(m);
  const size_t n = smt_.size();
  xyz_.resize(n * 3);
  const vecv &c = m.coordinates();
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) xyz_[3 * i + k] = c[i][k];
  float e = 0;
  ok(mi_vina_cache_eval_coords(v_.get(), xyz_.data(), smt_.data(), (int)n, 1, v, &e, nullptr));
  return e;
}

// cache::eval_deriv (cache.cpp:65-83): also leaves the gradient of every movable atom in m.minus_forces (0 for
// hydrogens)
fl HipCache::eval_deriv(model &m, fl v, const grid &user_grid) const {
  // cache::eval_deriv ignores its user_grid argument: the grid went into the lattice values at populate time
  // (cache.cpp:65-83,177-179).  Same here -- but only if this cache was built with it.
  if (user_grid.initialized() && !have_user_grid_)
    throw internal_error("HipCache was built without the user grid (pass its file text to the constructor)", 0);
  types_of(m);
  const size_t n = smt_.size();
  xyz_.resize(n * 3);
  forces_.resize(n * 3);
  const vecv &c = m.coordinates();
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) xyz_[3 * i + k] = c[i][k];
  float e = 0;
  ok(mi_vina_cache_eval_coords(v_.get(), xyz_.data(), smt_.data(), (int)n, 1, v, &e, forces_.data()));
  if (m.minus_forces.size() < n) m.minus_forces.resize(n);
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) m.minus_forces[i][k] = forces_[3 * i + k];
  return e;
}

void HipCache::set_ligand(const LigandArrays &lig) {
  mi_ligand_desc d = lig.desc();
  ok(mi_vina_set_ligand(v_.get(), &d));
  have_ligand_ = true;
}

fl HipCache::bfgs(std::vector<float> &conf, std::vector<float> &grad, const vec &v, unsigned maxiters) const {
  if (!have_ligand_) throw internal_error("HipCache::bfgs needs set_ligand first", 0);
  const float v3[3] = {v[0], v[1], v[2]};
  float e = 0;
  grad.resize(conf.size() - 1);
  ok(mi_vina_bfgs_batch(v_.get(), conf.data(), 1, v3, (int)maxiters, &e, grad.data(), nullptr));
  return e;
}

bool HipQuasiNewton::operator()(igrid &ig, std::vector<float> &conf, std::vector<float> &grad, const vec &v,
                                fl &energy) const {
  const HipCache *hc = dynamic_cast<const HipCache *>(&ig);  // quasi_newton.cpp:52-53's dispatch
  if (!hc || !hc->has_ligand()) return false;
  energy = hc->bfgs(conf, grad, v, params.maxiters);
  return true;
}

#ifdef MI_GNINA_WITH_GNINA_HEADERS
namespace {
// depth-first, children in order: the order branches_set_conf consumes the torsions (tree.h:293-311)
void walk(const tree<segment> &t, int parent, const frame &pframe, const model &m, LigandArrays &out) {
  const int k = (int)out.parent.size();
  out.parent.push_back(parent);
  out.abeg.push_back((int32_t)t.node.begin);
  out.aend.push_back((int32_t)t.node.end);
  // segment(origin, begin, end, axis_root, parent): relative_origin = origin - parent origin, relative_axis = axis,
  // both taken at the identity orientation the tree is built in (tree.h:190-203)
  const vec ro = t.node.get_origin() - pframe.get_origin();
  for (int d = 0; d < 3; d++) {
    out.rel_origin.push_back(ro[d]);
    out.rel_axis.push_back(t.node.axis[d]);
  }
  for (const auto &c : t.children) walk(c, k, t.node, m, out);
}
}  // namespace

LigandArrays ligand_arrays_from_model(const model &m) {
  if (m.ligands.size() != 1) throw internal_error("ligand_arrays_from_model: exactly one ligand expected", 0);
  const ligand &lig = m.ligands[0];
  if (lig.begin != 0) throw internal_error("ligand_arrays_from_model: the ligand must start at atom 0", 0);
  LigandArrays out;
  const atomv &atoms = m.get_movable_atoms();
  for (sz i = lig.begin; i < lig.end; i++) {
    out.smt.push_back((int32_t)atoms[i].sm);
    for (int d = 0; d < 3; d++) out.local_xyz.push_back(atoms[i].coords[d]);  // coordinates in the owning frame
  }
  out.parent.push_back(-1);
  out.abeg.push_back((int32_t)lig.node.begin);
  out.aend.push_back((int32_t)lig.node.end);
  for (int d = 0; d < 3; d++) {
    out.rel_origin.push_back(0.f);
    out.rel_axis.push_back(0.f);
  }
  for (const auto &c : lig.children) walk(c, 0, lig.node, m, out);
  for (const interacting_pair &p : lig.pairs) {
    out.pairs.push_back((int32_t)p.a);
    out.pairs.push_back((int32_t)p.b);
  }
  return out;
}

std::vector<float> flatten(const conf &c) {
  std::vector<float> x;
  const ligand_conf &l = c.ligands[0];
  for (int d = 0; d < 3; d++) x.push_back(l.rigid.position[d]);
  x.push_back(l.rigid.orientation.R_component_1());
  x.push_back(l.rigid.orientation.R_component_2());
  x.push_back(l.rigid.orientation.R_component_3());
  x.push_back(l.rigid.orientation.R_component_4());
  for (fl t : l.torsions) x.push_back(t);
  return x;
}
void unflatten(const std::vector<float> &x, conf &c) {
  ligand_conf &l = c.ligands[0];
  l.rigid.position = vec(x[0], x[1], x[2]);
  l.rigid.orientation = qt(x[3], x[4], x[5], x[6]);
  for (size_t i = 0; i < l.torsions.size(); i++) l.torsions[i] = x[7 + i];
}
void unflatten(const std::vector<float> &g, change &c) {
  ligand_change &l = c.ligands[0];
  l.rigid.position = vec(g[0], g[1], g[2]);
  l.rigid.orientation = vec(g[3], g[4], g[5]);
  for (size_t i = 0; i < l.torsions.size(); i++) l.torsions[i] = g[6 + i];
}

bool HipQuasiNewton::operator()(model &m, const precalculate &, igrid &ig, output_type &out, change &g, const vec &v,
                                const grid &user_grid) const {
  if (user_grid.initialized()) {
    auto *hc = dynamic_cast<HipCache *>(&ig);
    if (!hc || !hc->has_user_grid()) return false;
  }
  std::vector<float> x = flatten(out.c), grad;
  fl e = 0;
  if (!(*this)(ig, x, grad, v, e)) return false;
  unflatten(x, out.c);
  unflatten(grad, g);
  out.e = e;
  m.set(out.c);  // the model ends on the RETURNED conformation (what refine_structure's "m.set(out.c); // just to be sure"
                 // establishes anyway, main.cpp:152); the reference's CPU bfgs<> leaves it on the last trial it evaluated
  return true;
}
#endif

}  // namespace gnina_amd
