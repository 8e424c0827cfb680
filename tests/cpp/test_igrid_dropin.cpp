// The igrid seam with THE REFERENCE'S OWN CODE on both sides of it.
//
// Built only where /root/reference exists (oracle/Makefile.ref, target dropin): this file and
// gnina_amd/host/hip_cache.cpp are compiled against gnina's real headers (MI_GNINA_WITH_GNINA_HEADERS, behind
// oracle/ref_shims) and linked with the reference's objects of oracle/_ref and with libmi_gnina.so.  Then
//   1. gnina's parse_pdbqt.cpp builds the `model` from PDBQT text,
//   2. HipCache (HIP engine) and gnina's own `cache` are populated for the same box,
//   3. gnina's model::eval_deriv, quasi_newton (CPU bfgs<>) and monte_carlo run with HipCache as their igrid --
//      unchanged reference code calling the MI355X engine through igrid::eval / eval_deriv -- next to the same calls on
//      gnina's cache,
//   4. HipQuasiNewton hands whole minimisations to the device (the quasi_newton.cpp:52-72 style dispatch).
// Prints numbers; tests/test_host_adapter.py compares them (needs a GPU; the binary travels to the GPU box prebuilt).
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../gnina_amd/host/hip_cache.h"
#include "cache.h"
#include "custom_terms.h"
#include "monte_carlo.h"
#include "parse_pdbqt.h"
#include "precalculate.h"
#include "quasi_newton.h"
#include "weighted_terms.h"

model parse_ligand_stream_pdbqt(const std::string &name, std::istream &in);

static std::string slurp(const char *p) {
  std::ifstream f(p);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

static int run(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s rigid.pdbqt ligand.pdbqt\n", argv[0]);
    return 2;
  }
  if (mi_gnina_init(0) != MI_OK) {
    std::fprintf(stderr, "init failed: %s\n", mi_last_error());
    return 3;
  }
  std::istringstream rin(slurp(argv[1])), lin(slurp(argv[2]));
  model m = parse_receptor_pdbqt("rigid", rin);
  model lig = parse_ligand_stream_pdbqt("ligand", lin);
  m.append(lig);

  custom_terms t;   // main.cpp:1324-1329
  t.add("gauss(o=0,_w=0.5,_c=8)", -0.035579);
  t.add("gauss(o=3,_w=2,_c=8)", -0.005156);
  t.add("repulsion(o=0,_c=8)", 0.840245);
  t.add("hydrophobic(g=0.5,_b=1.5,_c=8)", -0.035069);
  t.add("non_dir_h_bond(g=-0.7,_b=0,_c=8)", -0.587439);
  t.add("num_tors_div", 5 * 0.05846 / 0.1 - 1);
  weighted_terms wt(&t, t.weights());
  precalculate_linear prec(wt, 32.0);

  grid_dims gd = m.movable_atoms_box(4.0, 0.375);   // --autobox_ligand with autobox_add 4 (main.cpp)
  std::vector<smt> need;
  m.get_movable_atom_types(need);
  grid user_grid;
  cache ref_cache("scoring_function_version001", gd, 1e3);
  ref_cache.populate(m, prec, need, user_grid, false);
  gnina_amd::HipCache hip(m, gd, need, 1e3);

  const vec v(1000, 1000, 1000), hunt(10, 10, 10);
  conf c0 = m.get_initial_conf(false);
  rng gen(7);
  // (1) model::eval_deriv through the igrid seam, reference cache vs HipCache
  for (int k = 0; k < 4; k++) {
    conf c = c0;
    if (k) c.randomize(vec(gd[0].begin, gd[1].begin, gd[2].begin), vec(gd[0].end, gd[1].end, gd[2].end), gen);
    change g1(m.get_size(), false), g2(m.get_size(), false);
    const fl e1 = m.eval_deriv(prec, ref_cache, v, c, g1, user_grid);
    const fl e2 = m.eval_deriv(prec, hip, v, c, g2, user_grid);
    double dmax = 0, gmax = 0;
    std::vector<float> a = gnina_amd::flatten(c);  // (just to size things)
    (void)a;
    for (sz i = 0; i < g1.num_floats(); i++) {
      dmax = std::max(dmax, (double)std::fabs(g1(i) - g2(i)));
      gmax = std::max(gmax, (double)std::fabs(g1(i)));
    }
    std::printf("eval_deriv %d %.9g %.9g dchange %.3g scale %.3g ig_eval %.9g %.9g\n", k, e1, e2, dmax, gmax,
                ref_cache.eval(m, v[1]), hip.eval(m, v[1]));
  }
  // (2) gnina's CPU quasi_newton minimising ON the HIP igrid vs on its own cache
  minimization_params mp;
  mp.maxiters = (25 + m.num_movable_atoms()) / 3;
  quasi_newton qn(mp);
  for (int k = 0; k < 3; k++) {
    conf c = c0;
    if (k) c.randomize(vec(gd[0].begin, gd[1].begin, gd[2].begin), vec(gd[0].end, gd[1].end, gd[2].end), gen);
    output_type o1(c, 0), o2(c, 0);
    change g(m.get_size(), false);
    qn(m, prec, ref_cache, o1, g, hunt, user_grid);
    qn(m, prec, hip, o2, g, hunt, user_grid);
    std::printf("cpu_bfgs_on %d ref_cache %.9g hip_cache %.9g\n", k, o1.e, o2.e);
    // (4) the same start, whole minimisation on the device
    gnina_amd::HipQuasiNewton hqn(mp);
    if (k == 0) hip.set_ligand(gnina_amd::ligand_arrays_from_model(m));
    output_type o3(c, 0);
    const bool took = hqn(m, prec, hip, o3, g, hunt, user_grid);
    std::printf("device_bfgs %d took %d e %.9g\n", k, (int)took, o3.e);
    const bool took_ref = hqn(m, prec, ref_cache, o3, g, hunt, user_grid);
    std::printf("device_bfgs_on_ref_cache took %d\n", (int)took_ref);
  }
  // (3) gnina's monte_carlo with the HIP igrid for both roles (search and Metropolis)
  monte_carlo mc;
  mc.num_steps = 30;
  mc.ssd_par.evals = mp.maxiters;
  mc.ssd_par.minparm = mp;
  mc.min_rmsd = 1.0;
  mc.num_saved_mins = 20;
  mc.hunt_cap = vec(10, 10, 10);
  for (int which = 0; which < 2; which++) {
    model mm = m;
    rng g2(42);
    output_container out;
    igrid &ig = which ? static_cast<igrid &>(hip) : static_cast<igrid &>(ref_cache);
    mc(mm, out, prec, ig, vec(gd[0].begin, gd[1].begin, gd[2].begin), vec(gd[0].end, gd[1].end, gd[2].end), NULL, g2,
       user_grid, ig);
    std::printf("mc_on %s n %zu best %.9g\n", which ? "hip_cache" : "ref_cache", out.size(), out[0].e);
  }
  return 0;
}

int main(int argc, char **argv) {
  try {
    return run(argc, argv);
  } catch (const internal_error &e) {  // gnina's own type: (file, line) -- HipCache puts its message into `file`
    fprintf(stderr, "internal_error: %s (%u)\n", e.file.c_str(), e.line);
  } catch (const std::exception &e) {
    fprintf(stderr, "exception: %s\n", e.what());
  }
  return 3;
}
