// oracle/_ref driver -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
//
// A small C ABI over the REFERENCE's own Vina/smina code: this file is ours, everything it calls is compiled from
// the unmodified sources under /root/reference/gninasrc/lib (parse_pdbqt.cpp, model.cpp, model.cu, tree.h, conf.h,
// quaternion.h, everything.cpp, weighted_terms.cpp, precalculate.h, cache.cpp, grid.cpp, non_cache.cpp,
// szv_grid.cpp, quasi_newton.cpp + bfgs.h, monte_carlo.cpp, mutate.cpp, coords.cpp, random.cpp ...) behind the
// stand-in headers of oracle/ref_shims/ (CUDA, Boost, OpenBabel are absent from this image).  The result,
// oracle/_ref/libgnina_ref.so, is what pins oracle/vina_ref.c and gnina_amd/host/pdbqt.cpp to the reference.
// It exists only where /root/reference does (this container); the GPU box uses the prebuilt file.
//
// Layout conventions shared with include/mi_gnina.h:
//   conf   = [position 3][orientation quaternion a,b,c,d][ligand torsions T_lig][flex torsions, residue by residue]
//   change = [force 3][torque 3][ligand torsion derivatives][flex torsion derivatives]
#include <cxxabi.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>

#include "cache.h"
#include "coords.h"
#include "custom_terms.h"
#include "everything.h"
#include "model.h"
#include "monte_carlo.h"
#include "mutate.h"
#include "naive_non_cache.h"
#include "non_cache.h"
#include "parse_error.h"
#include "parse_pdbqt.h"
#include "precalculate.h"
#include "quasi_newton.h"
#include "weighted_terms.h"

model parse_ligand_stream_pdbqt(const std::string &name, std::istream &in);  // parse_pdbqt.cpp:527

// model befriends `model_test` (model.h): the legitimate door to its private members.
struct model_test {
  static const interacting_pairs &ligand_pairs(const model &m) { return m.ligands[0].pairs; }
  static sz n_flex(const model &m) { return m.flex.size(); }
  static const vecv &internal_coords(const model &m) { return m.internal_coords; }
  static const atom &get_atom(const model &m, sz i) { return m.get_atom(m.sz_to_atom_index(i)); }
  static fl pairs_deriv(const model &m, const precalculate &p, fl v, const interacting_pairs &pairs, vecv &forces) {
    return m.eval_interacting_pairs_deriv(p, v, pairs, m.coords, forces);
  }
};

namespace {
thread_local std::string g_err;

struct Scene {
  model m;
  model m0;  // as parsed: every Monte-Carlo chain starts from a copy of it (parallel_mc.cpp:66-70 copies the model per task)
  custom_terms t;
  std::unique_ptr<weighted_terms> wt;
  std::unique_ptr<precalculate_linear> prec;
  std::unique_ptr<precalculate_exact> exact;
  std::unique_ptr<precalculate_splines> splines;
  std::unique_ptr<precalculate_splines> run_splines;  // --approximation spline: the run's precalculate (ref_set_approximation)
  grid_dims gd;
  grid user_grid;
  std::unique_ptr<cache> c;
  std::unique_ptr<szv_grid_cache> gridcache;
  std::unique_ptr<non_cache> nc;
  bool has_ligand = false;
  bool accurate_ls = false;  // --accurate_line_search (minimization_params::BFGSAccurateLineSearch)
  bool simple = false;       // --simple_ascent (minimization_params::Simple)
};

conf to_conf(const Scene &s, const float *x) {
  conf c(s.m.get_size(), false);
  sz k = 0;
  VINA_FOR_IN(i, c.ligands) {
    ligand_conf &l = c.ligands[i];
    l.rigid.position = vec(x[k], x[k + 1], x[k + 2]);
    l.rigid.orientation = qt(x[k + 3], x[k + 4], x[k + 5], x[k + 6]);
    k += 7;
    VINA_FOR_IN(j, l.torsions) l.torsions[j] = x[k++];
  }
  VINA_FOR_IN(i, c.flex)
  VINA_FOR_IN(j, c.flex[i].torsions) c.flex[i].torsions[j] = x[k++];
  return c;
}
void from_conf(const conf &c, float *x) {
  sz k = 0;
  VINA_FOR_IN(i, c.ligands) {
    const ligand_conf &l = c.ligands[i];
    VINA_FOR(d, 3) x[k++] = l.rigid.position[d];
    x[k++] = l.rigid.orientation.R_component_1();
    x[k++] = l.rigid.orientation.R_component_2();
    x[k++] = l.rigid.orientation.R_component_3();
    x[k++] = l.rigid.orientation.R_component_4();
    VINA_FOR_IN(j, l.torsions) x[k++] = l.torsions[j];
  }
  VINA_FOR_IN(i, c.flex)
  VINA_FOR_IN(j, c.flex[i].torsions) x[k++] = c.flex[i].torsions[j];
}
void from_change(const change &g, float *x) {
  sz k = 0;
  VINA_FOR_IN(i, g.ligands) {
    VINA_FOR(d, 3) x[k++] = g.ligands[i].rigid.position[d];
    VINA_FOR(d, 3) x[k++] = g.ligands[i].rigid.orientation[d];
    VINA_FOR_IN(j, g.ligands[i].torsions) x[k++] = g.ligands[i].torsions[j];
  }
  VINA_FOR_IN(i, g.flex)
  VINA_FOR_IN(j, g.flex[i].torsions) x[k++] = g.flex[i].torsions[j];
}
void copy_coords(const model &m, float *xyz) {
  if (!xyz) return;
  VINA_FOR_IN(i, m.coords)
  VINA_FOR(d, 3) xyz[3 * i + d] = m.coords[i][d];
}
igrid &pick_ig(Scene &s, int which) {
  if (which == 0) {
    if (!s.c) throw std::runtime_error("cache not built");
    return *s.c;
  }
  if (!s.nc) throw std::runtime_error("non_cache not built");
  return *s.nc;
}
// the run's precalculate (main.cpp:1384-1391): linear unless ref_set_approximation chose splines
const precalculate &run_prec(Scene &s) {
  if (s.run_splines) return *s.run_splines;
  return *s.prec;
}
const precalculate &pick_prec(Scene &s, int which) {
  if (which == 1) return *s.exact;
  if (which == 2) {
    if (!s.splines) s.splines.reset(new precalculate_splines(*s.wt, 10.0));
    return *s.splines;
  }
  return run_prec(s);
}
}  // namespace

#define RTRY try {
#define RCATCH(ret)                                              \
  }                                                              \
  catch (const parse_error &e) {                                 \
    std::ostringstream o;                                        \
    o << "parse_error " << e.file << ":" << e.line << ": " << e.reason; \
    g_err = o.str();                                             \
    return ret;                                                  \
  }                                                              \
  catch (const internal_error &e) {                              \
    std::ostringstream o;                                        \
    o << "internal_error " << e.file << ":" << e.line;           \
    g_err = o.str();                                             \
    return ret;                                                  \
  }                                                              \
  catch (const std::exception &e) {                              \
    g_err = e.what();                                            \
    return ret;                                                  \
  }                                                              \
  catch (...) {                                                  \
    g_err = std::string("exception of type ") + abi::__cxa_current_exception_type()->name(); \
    return ret;                                                  \
  }

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }

// main.cpp:1324-1329 default terms, or six weights in that order (gauss1, gauss2, repulsion, hydrophobic,
// non_dir_h_bond, num_tors_div).  rigid / flex / ligand are PDBQT texts; flex and ligand may be NULL.
// The model is assembled as molgetter.cpp does: parse_receptor_pdbqt(rigid[, flex]) then m.append(ligand).
void *ref_scene_create(const char *rigid, const char *flex, const char *ligand, const float *weights) {
  RTRY
  std::unique_ptr<Scene> s(new Scene);
  std::istringstream rin(rigid ? rigid : "");
  if (flex && *flex) {
    std::istringstream fin(flex);
    s->m = parse_receptor_pdbqt("rigid", rin, "flex", fin);
  } else {
    s->m = parse_receptor_pdbqt("rigid", rin);
  }
  if (ligand && *ligand) {
    std::istringstream lin(ligand);
    model lig = parse_ligand_stream_pdbqt("ligand", lin);
    s->m.append(lig);
    s->has_ligand = true;
  }
  s->m0 = s->m;
  const float dflt[6] = {-0.035579f, -0.005156f, 0.840245f, -0.035069f, -0.587439f, (float)(5 * 0.05846 / 0.1 - 1)};
  const float *w = weights ? weights : dflt;
  s->t.add("gauss(o=0,_w=0.5,_c=8)", w[0]);
  s->t.add("gauss(o=3,_w=2,_c=8)", w[1]);
  s->t.add("repulsion(o=0,_c=8)", w[2]);
  s->t.add("hydrophobic(g=0.5,_b=1.5,_c=8)", w[3]);
  s->t.add("non_dir_h_bond(g=-0.7,_b=0,_c=8)", w[4]);
  s->t.add("num_tors_div", w[5]);
  s->wt.reset(new weighted_terms(&s->t, s->t.weights()));
  s->prec.reset(new precalculate_linear(*s->wt, 32.0));
  s->exact.reset(new precalculate_exact(*s->wt));
  return s.release();
  RCATCH(nullptr)
}
void ref_scene_free(void *h) { delete (Scene *)h; }

// sizes: [0] movable+inflex atoms (model::atoms), [1] num_movable_atoms, [2] grid_atoms, [3] ligand torsions,
// [4] flex residues, [5] flex torsions, [6] ligand pairs, [7] other pairs, [8] ligand begin, [9] ligand end
int ref_sizes(void *h, int *out) {
  RTRY
  Scene &s = *(Scene *)h;
  const model &m = s.m;
  conf_size cs = m.get_size();
  out[0] = (int)m.atoms.size();
  out[1] = (int)m.num_movable_atoms();
  out[2] = (int)m.grid_atoms.size();
  out[3] = cs.ligands.empty() ? 0 : (int)cs.ligands[0];
  out[4] = (int)cs.flex.size();
  int ft = 0;
  VINA_FOR_IN(i, cs.flex) ft += (int)cs.flex[i];
  out[5] = ft;
  out[6] = m.ligands.size() ? (int)m.ligands[0].pairs.size() : 0;
  out[7] = (int)m.other_pairs.size();
  out[8] = m.ligands.size() ? (int)m.ligands[0].begin : 0;
  out[9] = m.ligands.size() ? (int)m.ligands[0].end : 0;
  return 0;
  RCATCH(1)
}
// model::atoms (movable, then inflex): current coordinates, smina types after assign_types, internal coordinates
int ref_atoms(void *h, float *xyz, int *smt, float *internal) {
  RTRY
  const model &m = ((Scene *)h)->m;
  VINA_FOR_IN(i, m.atoms) {
    VINA_FOR(d, 3) {
      if (xyz) xyz[3 * i + d] = m.coords[i][d];
      if (internal) internal[3 * i + d] = model_test::internal_coords(m)[i][d];
    }
    if (smt) smt[i] = (int)m.atoms[i].sm;
  }
  return 0;
  RCATCH(1)
}
int ref_grid_atoms(void *h, float *xyz, int *smt) {
  RTRY
  const model &m = ((Scene *)h)->m;
  VINA_FOR_IN(i, m.grid_atoms) {
    VINA_FOR(d, 3) xyz[3 * i + d] = m.grid_atoms[i].coords[d];
    smt[i] = (int)m.grid_atoms[i].sm;
  }
  return 0;
  RCATCH(1)
}
// which = 0: ligands[0].pairs, 1: other_pairs.  out [n][2] atom indices, t12 [n][2] the two smina types
int ref_pairs(void *h, int which, int *out, int *t12) {
  RTRY
  const model &m = ((Scene *)h)->m;
  const interacting_pairs &p = which == 0 ? model_test::ligand_pairs(m) : m.other_pairs;
  VINA_FOR_IN(i, p) {
    out[2 * i] = (int)p[i].a;
    out[2 * i + 1] = (int)p[i].b;
    if (t12) {
      t12[2 * i] = (int)p[i].t1;
      t12[2 * i + 1] = (int)p[i].t2;
    }
  }
  return 0;
  RCATCH(1)
}
// bonds of atom i in the combined index space (grid_atoms first, then atoms), in bond-list order
int ref_bonds(void *h, int i, int *out, int cap) {
  RTRY
  const model &m = ((Scene *)h)->m;
  const atom &a = model_test::get_atom(m, (sz)i);
  int n = 0;
  VINA_FOR_IN(k, a.bonds) {
    if (n < cap) out[n] = (int)(a.bonds[k].connected_atom_index.in_grid ? a.bonds[k].connected_atom_index.i
                                                                         : m.grid_atoms.size() + a.bonds[k].connected_atom_index.i);
    n++;
  }
  return n;
  RCATCH(-1)
}
// precalculate_linear tables of a type pair (precalculate.h:165-272): n = 2051 points of fast / smooth (E, dor)
int ref_table_n(void *h) {
  Scene &s = *(Scene *)h;
  return (int)(s.prec->cutoff_sqr() * 32.0) + 3;
}
int ref_table_eval(void *h, int t1, int t2, const float *r2, int n, float *fast, float *e, float *dor) {
  RTRY
  Scene &s = *(Scene *)h;
  atom_base a, b;
  a.sm = (smt)t1;
  b.sm = (smt)t2;
  for (int i = 0; i < n; i++) {
    if (fast) fast[i] = s.prec->eval_fast((smt)t1, (smt)t2, r2[i]).eval(a, b);
    if (e || dor) {
      pr p = s.prec->eval_deriv(a, b, r2[i]);
      if (e) e[i] = p.first;
      if (dor) dor[i] = p.second;
    }
  }
  return 0;
  RCATCH(1)
}
// precalculate_exact / the scoring function itself at distance r (weighted_terms::eval_fast)
float ref_pair_energy(void *h, int t1, int t2, float r) {
  Scene &s = *(Scene *)h;
  atom_base a, b;
  a.sm = (smt)t1;
  b.sm = (smt)t2;
  return s.wt->eval_fast((smt)t1, (smt)t2, r).eval(a, b);
}
// setup_grid_dims (main.cpp:622-634) and the igrids: cache::populate for the movable atom types, non_cache
int ref_build_grids(void *h, const float *center, const float *size, float slope, int build_cache, float *begin,
                    float *end, int *n, const int *types, int n_types) {
  RTRY
  Scene &s = *(Scene *)h;
  const fl granularity = 0.375;
  VINA_FOR(i, 3) {
    s.gd[i].n = sz(std::ceil(size[i] / granularity));
    fl real_span = granularity * s.gd[i].n;
    s.gd[i].begin = center[i] - real_span / 2;
    s.gd[i].end = s.gd[i].begin + real_span;
    begin[i] = s.gd[i].begin;
    end[i] = s.gd[i].end;
    n[i] = (int)s.gd[i].n;
  }
  s.gridcache.reset(new szv_grid_cache(s.m, run_prec(s).cutoff_sqr()));
  s.nc.reset(new non_cache(*s.gridcache, s.gd, &run_prec(s), slope));
  if (build_cache) {
    s.c.reset(new cache("scoring_function_version001", s.gd, slope));
    std::vector<smt> need;
    s.m.get_movable_atom_types(need);
    for (int i = 0; i < n_types; i++)
      if (!has(need, (smt)types[i])) need.push_back((smt)types[i]);
    s.c->populate(s.m, run_prec(s), need, s.user_grid, false);
  }
  return 0;
  RCATCH(1)
}
}  // extern "C"

extern "C" {
// --approximation spline (main.cpp:1386-1387): the run's precalculate becomes precalculate_splines(wt, factor); kind 0
// returns to precalculate_linear(wt, 32).  Before ref_build_grids.
int ref_set_approximation(void *h, int kind, float factor) {
  RTRY
  Scene &s = *(Scene *)h;
  if (kind == 1) s.run_splines.reset(new precalculate_splines(*s.wt, factor));
  else s.run_splines.reset();
  return 0;
  RCATCH(1)
}
int ref_set_line_search(void *h, int kind) {
  RTRY
  ((Scene *)h)->accurate_ls = kind == 1;
  ((Scene *)h)->simple = kind == 2;
  return 0;
  RCATCH(1)
}
// one spline's value and derivative: precalculate_splines::eval_fast / eval_deriv through the run's precalculate
int ref_prec_eval(void *h, int t1, int t2, const float *r2, int n, float *e, float *dor) {
  RTRY
  Scene &s = *(Scene *)h;
  atom a, b;
  a.sm = (smt)t1;
  b.sm = (smt)t2;
  for (int i = 0; i < n; i++) {
    pr p = run_prec(s).eval_deriv(a, b, r2[i]);
    e[i] = p.first;
    dor[i] = p.second;
  }
  return 0;
  RCATCH(1)
}
}  // extern "C"

extern "C" {
// --user_grid: grid::init(gd, user_in, ug_scaling_factor) (grid.cpp:69-92) on the value lines of the file; the grid_dims
// come from the caller (setup_user_gd lives in main.cpp, which this library does not compile).  Before ref_build_grids.
int ref_set_user_grid(void *h, const float *begin, const float *end, const int *n, const char *value_lines, float scale) {
  RTRY
  Scene &s = *(Scene *)h;
  if (!value_lines) {
    s.user_grid = grid();
    return 0;
  }
  grid_dims ugd;
  VINA_FOR(i, 3) {
    ugd[i].begin = begin[i];
    ugd[i].end = end[i];
    ugd[i].n = (sz)n[i];
  }
  std::istringstream in(value_lines);
  s.user_grid.init(ugd, in, scale);
  return 0;
  RCATCH(1)
}
}  // extern "C"

// cache::grids is private and cache has no test friend: read a grid back through the public evaluation instead --
// at a lattice point trilinear interpolation returns the stored value exactly (weights 1 and 0).
extern "C" {
// energies of a one-atom probe of type t at the given points: grid::evaluate through cache::eval
int ref_cache_probe(void *h, int t, const float *xyz, int n, float v, float *out_e, float *out_deriv) {
  RTRY
  Scene &s = *(Scene *)h;
  if (!s.c) throw std::runtime_error("cache not built");
  model probe;
  probe.atoms.resize(1);
  probe.atoms[0].sm = (smt)t;
  probe.coords.resize(1);
  probe.minus_forces.resize(1);
  probe.m_num_movable_atoms = 1;
  for (int i = 0; i < n; i++) {
    probe.coords[0] = vec(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    if (out_deriv) {
      out_e[i] = s.c->eval_deriv(probe, v, s.user_grid);
      VINA_FOR(d, 3) out_deriv[3 * i + d] = probe.minus_forces[0][d];
    } else {
      out_e[i] = s.c->eval(probe, v);
    }
  }
  return 0;
  RCATCH(1)
}

// model::set(conf) -> coordinates of model::atoms
int ref_set_conf(void *h, const float *x, float *xyz) {
  RTRY
  Scene &s = *(Scene *)h;
  s.m.set(to_conf(s, x));
  copy_coords(s.m, xyz);
  return 0;
  RCATCH(1)
}
int ref_initial_conf(void *h, float *x) {
  RTRY
  Scene &s = *(Scene *)h;
  from_conf(s.m.get_initial_conf(false), x);
  return 0;
  RCATCH(1)
}
// model::eval_deriv (model.cu:202-225).  ig: 0 cache, 1 non_cache.  prec: 0 linear, 1 exact, 2 splines.
int ref_eval_deriv(void *h, const float *x, const float *v3, int ig, int prec, float *energy, float *chg, float *xyz,
                   float *minus_forces) {
  RTRY
  Scene &s = *(Scene *)h;
  conf c = to_conf(s, x);
  change g(s.m.get_size(), false);
  *energy = s.m.eval_deriv(pick_prec(s, prec), pick_ig(s, ig), vec(v3[0], v3[1], v3[2]), c, g, s.user_grid);
  if (chg) from_change(g, chg);
  copy_coords(s.m, xyz);
  if (minus_forces) VINA_FOR_IN(i, s.m.minus_forces) VINA_FOR(d, 3) minus_forces[3 * i + d] = s.m.minus_forces[i][d];
  return 0;
  RCATCH(1)
}
// model::eval (model.cu:227-233): the Metropolis energy of monte_carlo.cpp:44-47 is ig.eval after m.set
int ref_eval(void *h, const float *x, const float *v3, int ig, int prec, float *energy) {
  RTRY
  Scene &s = *(Scene *)h;
  *energy = s.m.eval(pick_prec(s, prec), pick_ig(s, ig), vec(v3[0], v3[1], v3[2]), to_conf(s, x), s.user_grid);
  return 0;
  RCATCH(1)
}
// igrid::eval alone (receptor term) after model::set
int ref_ig_eval(void *h, const float *x, float v, int ig, float *energy) {
  RTRY
  Scene &s = *(Scene *)h;
  s.m.set(to_conf(s, x));
  *energy = pick_ig(s, ig).eval(s.m, v);
  return 0;
  RCATCH(1)
}
// do_search's reported energies (main.cpp:339-344): intramolecular = eval_intramolecular(exact_prec),
// e = eval_adjusted(sf, exact_prec, non_cache, authentic_v, conf, intramolecular)
int ref_final_energies(void *h, const float *x, const float *v3, float *e_final, float *intramolecular) {
  RTRY
  Scene &s = *(Scene *)h;
  conf c = to_conf(s, x);
  const vec v(v3[0], v3[1], v3[2]);
  const fl intra = s.m.eval_intramolecular(*s.exact, v, c);
  *e_final = s.m.eval_adjusted(*s.wt, *s.exact, pick_ig(s, 1), v, c, intra, s.user_grid);
  *intramolecular = intra;
  return 0;
  RCATCH(1)
}
// quasi_newton::operator() (quasi_newton.cpp:49-83): bfgs<> + fast_line_search, in place
int ref_bfgs(void *h, float *x, const float *v3, int ig, int max_iters, float *energy, float *chg) {
  RTRY
  Scene &s = *(Scene *)h;
  minimization_params mp;
  mp.maxiters = (unsigned)max_iters;
  if (s.accurate_ls) mp.type = minimization_params::BFGSAccurateLineSearch;
  if (s.simple) mp.type = minimization_params::Simple;
  quasi_newton qn(mp);
  output_type out(to_conf(s, x), 0);
  change g(s.m.get_size(), false);
  qn(s.m, run_prec(s), pick_ig(s, ig), out, g, vec(v3[0], v3[1], v3[2]), s.user_grid);
  from_conf(out.c, x);
  *energy = out.e;
  if (chg) from_change(g, chg);
  return 0;
  RCATCH(1)
}
// conf::increment(change, alpha) (conf.h:409-419): the line-search step
int ref_conf_increment(void *h, float *x, const float *chg, float alpha) {
  RTRY
  Scene &s = *(Scene *)h;
  conf c = to_conf(s, x);
  change g(s.m.get_size(), false);
  sz k = 0;
  VINA_FOR_IN(i, g.ligands) {
    g.ligands[i].rigid.position = vec(chg[k], chg[k + 1], chg[k + 2]);
    g.ligands[i].rigid.orientation = vec(chg[k + 3], chg[k + 4], chg[k + 5]);
    k += 6;
    VINA_FOR_IN(j, g.ligands[i].torsions) g.ligands[i].torsions[j] = chg[k++];
  }
  VINA_FOR_IN(i, g.flex)
  VINA_FOR_IN(j, g.flex[i].torsions) g.flex[i].torsions[j] = chg[k++];
  c.increment(g, alpha);
  from_conf(c, x);
  return 0;
  RCATCH(1)
}
// monte_carlo::operator() (monte_carlo.cpp:99-148) for one chain with gnina's settings (main.cpp:441-463).
// The random stream comes from oracle/ref_shims/boost/random.hpp (mt19937 exact; distributions restated).
// Returns the number of saved poses; e [num_saved], confs [num_saved][conf_len], coords [num_saved][n_heavy][3].
int ref_mc(void *h, unsigned seed, int n_steps, int max_iters, int num_saved, float temperature, float min_rmsd,
           const float *corner1, const float *corner2, int ig, int conf_len, int n_heavy, float *e, float *confs,
           float *coords) {
  RTRY
  Scene &s = *(Scene *)h;
  monte_carlo mc;
  mc.num_steps = (unsigned)n_steps;
  mc.temperature = temperature;
  mc.ssd_par.evals = (unsigned)max_iters;
  mc.ssd_par.minparm.maxiters = (unsigned)max_iters;
  if (s.accurate_ls) mc.ssd_par.minparm.type = minimization_params::BFGSAccurateLineSearch;
  if (s.simple) mc.ssd_par.minparm.type = minimization_params::Simple;
  mc.min_rmsd = min_rmsd;
  mc.num_saved_mins = (sz)num_saved;
  mc.hunt_cap = vec(10, 10, 10);
  rng generator(static_cast<rng::result_type>(seed));
  output_container out;
  igrid &g = pick_ig(s, ig);
  s.m = s.m0;
  mc(s.m, out, run_prec(s), g, vec(corner1[0], corner1[1], corner1[2]), vec(corner2[0], corner2[1], corner2[2]), NULL,
     generator, s.user_grid, g);
  int n = 0;
  VINA_FOR_IN(i, out) {
    if ((int)i >= num_saved) break;
    e[i] = out[i].e;
    from_conf(out[i].c, confs + (size_t)i * conf_len);
    for (int a = 0; a < n_heavy && a < (int)out[i].coords.size(); a++)
      VINA_FOR(d, 3) coords[((size_t)i * n_heavy + a) * 3 + d] = out[i].coords[a][d];
    n++;
  }
  return n;
  RCATCH(-1)
}
// parallel_mc's threading model (parallel_mc.cpp:183-214, main.cpp:1418-1442): n_chains Monte-Carlo tasks, each with its
// own copy of the model and its own generator, over ONE shared precalculate and igrid (both const), handed to
// n_threads workers.  bench.py's C3 CPU baseline: returns the wall-clock seconds of the whole fan-out in *seconds
// and every chain's best energy in best_e [n_chains] (so the work cannot be optimised away).
int ref_mc_parallel(void *h, const unsigned *seeds, int n_chains, int n_threads, int n_steps, int max_iters,
                    int num_saved, const float *corner1, const float *corner2, int ig, double *seconds, float *best_e) {
  RTRY
  Scene &s = *(Scene *)h;
  igrid &g = pick_ig(s, ig);
  const precalculate &p = run_prec(s);
  const vec c1(corner1[0], corner1[1], corner1[2]), c2(corner2[0], corner2[1], corner2[2]);
  std::atomic<int> next(0);
  std::atomic<int> failed(0);
  auto worker = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n_chains) return;
      try {
        monte_carlo mc;
        mc.num_steps = (unsigned)n_steps;
        mc.temperature = 1.2;
        mc.ssd_par.evals = (unsigned)max_iters;
        mc.ssd_par.minparm.maxiters = (unsigned)max_iters;
        mc.min_rmsd = 1.0;
        mc.num_saved_mins = (sz)num_saved;
        mc.hunt_cap = vec(10, 10, 10);
        rng generator(static_cast<rng::result_type>(seeds[i]));
        output_container out;
        model m = s.m0;  // parallel_mc_task holds its own model (parallel_mc.cpp:66-70)
        mc(m, out, p, g, c1, c2, NULL, generator, s.user_grid, g);
        best_e[i] = out.empty() ? 0.f : (float)out[0].e;
      } catch (...) {
        failed++;
      }
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; t++) pool.emplace_back(worker);
  for (auto &t : pool) t.join();
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (failed) throw std::runtime_error("a Monte-Carlo task threw");
  return 0;
  RCATCH(1)
}
// mutate_conf (mutate.cpp:35-73) with a seeded generator: one mutation of x, in place
int ref_mutate(void *h, float *x, unsigned seed, float amplitude) {
  RTRY
  Scene &s = *(Scene *)h;
  conf c = to_conf(s, x);
  s.m.set(c);
  rng generator(static_cast<rng::result_type>(seed));
  mutate_conf(c, s.m, amplitude, generator);
  from_conf(c, x);
  return 0;
  RCATCH(1)
}
// non_cache::within (non_cache.cpp:181-199)
int ref_within(void *h, const float *x) {
  RTRY
  Scene &s = *(Scene *)h;
  s.m.set(to_conf(s, x));
  return s.nc->within(s.m) ? 1 : 0;
  RCATCH(-1)
}
// weighted_terms::conf_independent (the num_tors_div adjustment, everything.h:796-814)
float ref_conf_independent(void *h, float e) {
  Scene &s = *(Scene *)h;
  return s.wt->conf_independent(s.m, e);
}
// raw draws of the stand-in distributions, for tests that mirror the stream
int ref_random_stream(unsigned seed, int n, float *uniform01, int *ints_0_9, float *normals) {
  rng g(static_cast<rng::result_type>(seed));
  for (int i = 0; i < n; i++) uniform01[i] = random_fl(0, 1, g);
  for (int i = 0; i < n; i++) ints_0_9[i] = random_int(0, 9, g);
  for (int i = 0; i < n; i++) normals[i] = random_normal(0, 1, g);
  return 0;
}
}  // extern "C"
