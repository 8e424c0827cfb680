// The PRIMARY seam (DLScorer) with THE REFERENCE'S OWN CODE on both sides of it.
//
// Built only where /root/reference exists (oracle/Makefile.ref, target dropin_cnn): this file and
// gnina_amd/host/hip_cnn_scorer.cpp are compiled against gnina's REAL headers (MI_GNINA_WITH_GNINA_HEADERS: dl_scorer.h,
// model.h, user_opts.h behind oracle/ref_shims) and linked with the reference's own objects -- dl_scorer.cpp
// (DLScorer::setLigand / setReceptor / set_center_from_model, SURVEY 8 rows a1-a2, unmodified), non_cache_cnn.cpp
// (row a10), quasi_newton.cpp + bfgs.h, model.cpp / model.cu, parse_pdbqt.cpp -- and with libmi_gnina.so.  Then
//   1. gnina's parse_pdbqt.cpp builds the `model` from PDBQT text,
//   2. gnina_amd::HipCNNScorer : DLScorer (gnina's class) is constructed from gnina's cnn_options,
//   3. gnina's non_cache_cnn takes it as its DLScorer& and gnina's code runs on it unchanged:
//        non_cache_cnn::eval / eval_deriv (non_cache_cnn.cpp:33-169), model::eval_deriv through the igrid seam
//        (model.cu:202-225: the CNN gradient folded into change by gnina's own tree code), and refine_structure's loop
//        (main.cpp:131-171, restated below: main.cpp itself needs the CLI stack) with gnina's quasi_newton,
//   4. next to each number: the batched C-ABI entry points on the same conformations (mi_cnn_eval_batch,
//      mi_cnn_refine_batch), which is what a batched caller uses instead.
// Prints numbers; tests/test_host_adapter.py compares them (needs a GPU; the binary travels to the GPU box prebuilt).
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../gnina_amd/host/hip_cache.h"
#include "../../gnina_amd/host/hip_cnn_scorer.h"
#include "custom_terms.h"
#include "non_cache_cnn.h"
#include "parse_pdbqt.h"
#include "precalculate.h"
#include "quasi_newton.h"
#include "szv_grid.h"
#include "weighted_terms.h"

model parse_ligand_stream_pdbqt(const std::string &name, std::istream &in);

static std::string slurp(const char *p) {
  std::ifstream f(p);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

#define CK(call)                                                         \
  do {                                                                   \
    if ((call) != MI_OK) {                                               \
      std::fprintf(stderr, "%s failed: %s\n", #call, mi_last_error());   \
      return 4;                                                          \
    }                                                                    \
  } while (0)

static int run(int argc, char **argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s rigid.pdbqt ligand.pdbqt weights_dir [model_name]\n", argv[0]);
    return 2;
  }
  const std::string model_name = argc > 4 ? argv[4] : "default2017";
  if (mi_gnina_init(0) != MI_OK) {
    std::fprintf(stderr, "init failed: %s\n", mi_last_error());
    return 3;
  }
  gnina_amd::set_builtin_model_dir(argv[3]);
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

  // gnina's objects, as main.cpp:1427-1430 / 466-476 build them
  cnn_options opts;
  opts.cnn_model_names.push_back(model_name);
  opts.cnn_scoring = CNNrefinement;
  gnina_amd::HipCNNScorer cnn(opts);
  if (!cnn.initialized()) {
    std::fprintf(stderr, "scorer not initialised\n");
    return 3;
  }
  grid_dims gd = m.movable_atoms_box(4.0, 0.375);
  szv_grid_cache gridcache(m, prec.cutoff_sqr());
  const fl slope = 1e3;
  non_cache_cnn nc(gridcache, gd, &prec, slope, cnn);   // <- gnina's igrid over OUR DLScorer
  grid user_grid;
  const vec v(1000, 1000, 1000);

  // the batched C-ABI objects for the same complex
  const std::string blob = std::string(argv[3]) + "/" + model_name + ".mgw";
  mi_model *mm = mi_model_load_file(blob.c_str());
  if (!mm) {
    std::fprintf(stderr, "mi_model_load_file: %s\n", mi_last_error());
    return 4;
  }
  mi_scorer *sc = mi_scorer_create(&mm, 1);
  {
    std::vector<float> xyz;
    std::vector<int32_t> ty;
    for (const atom &a : m.get_fixed_atoms()) {   // DLScorer::setReceptor's rows without flexible residues: the fixed atoms
      xyz.push_back(a.coords[0]), xyz.push_back(a.coords[1]), xyz.push_back(a.coords[2]);
      ty.push_back((int32_t)a.sm);
    }
    CK(mi_scorer_set_receptor(sc, xyz.data(), ty.data(), (int)ty.size()));
  }
  mi_vina *vina = mi_vina_create(nullptr, 8.f, 32.f);
  gnina_amd::LigandArrays la = gnina_amd::ligand_arrays_from_model(m);
  mi_ligand_desc desc = la.desc();
  CK(mi_vina_set_ligand(vina, &desc));
  mi_cnn_box box;
  std::memset(&box, 0, sizeof box);
  box.use_search_box = 1;
  for (int k = 0; k < 3; k++) box.box_begin[k] = gd[k].begin, box.box_end[k] = gd[k].end;
  box.cnn_dimension = cnn.get_grid_dim();
  box.slope = slope;
  box.empirical_weight = 1.f;
  box.v = 1000.f;

  conf c0 = m.get_initial_conf(false);
  rng gen(11);
  const int n = (int)(6 + la.n_torsions());
  // (1) non_cache_cnn::eval / eval_deriv and model::eval_deriv through the seam, vs mi_cnn_eval_batch
  for (int k = 0; k < 4; k++) {
    conf c = c0;
    if (k) {  // gentle random moves about the crystal pose (stay inside both boxes most of the time)
      c.ligands[0].rigid.position += vec(random_fl(-1, 1, gen), random_fl(-1, 1, gen), random_fl(-1, 1, gen));
      VINA_FOR_IN(j, c.ligands[0].torsions) c.ligands[0].torsions[j] += random_fl(-0.5, 0.5, gen);
      if (k == 3) c.ligands[0].rigid.position += vec(9, 0, 0);   // partly outside the search box: penalties active
    }
    m.set(c);
    const fl e_eval = nc.eval(m, v[1]);
    change g(m.get_size(), false);
    const fl e_deriv = m.eval_deriv(prec, nc, v, c, g, user_grid);
    std::vector<float> x = gnina_amd::flatten(c), ch(n), e1(1), e0(1);
    CK(mi_cnn_eval_batch(vina, sc, x.data(), 1, &box, nullptr, 1, e1.data(), ch.data()));
    CK(mi_cnn_eval_batch(vina, sc, x.data(), 1, &box, nullptr, 0, e0.data(), nullptr));
    double dmax = 0, gmax = 0;
    for (int i = 0; i < n; i++) {
      dmax = std::max(dmax, (double)std::fabs(g(i) - ch[i]));
      gmax = std::max(gmax, (double)std::fabs(g(i)));
    }
    std::printf("cnn_eval %d gnina_eval %.9g abi_eval %.9g gnina_eval_deriv %.9g abi_eval_deriv %.9g dchange %.3g scale %.3g\n",
                k, e_eval, e0[0], e_deriv, e1[0], dmax, gmax);
  }
  // (2) DLScorer::score through gnina's own setLigand / setReceptor vs mi_scorer_score_batch on the same coordinates
  {
    m.set(c0);
    float aff = 0, loss = 0, var = 0;
    const float pose = cnn.score(m, false, aff, loss, var);
    std::vector<float> xyz;
    std::vector<int32_t> ty;
    const sz off = 0;   // the ligand's atoms start the model here (no flexible residues)
    for (sz i = off; i < m.num_movable_atoms(); i++) {
      const vec &p = m.coordinates()[i];
      xyz.push_back(p[0]), xyz.push_back(p[1]), xyz.push_back(p[2]);
      ty.push_back((int32_t)m.get_movable_atoms()[i].sm);
    }
    float p2, a2, l2;
    CK(mi_scorer_score_batch(sc, xyz.data(), ty.data(), 1, (int)ty.size(), nullptr, &p2, &a2, &l2, nullptr));
    std::printf("score gnina_seam %.9g %.9g %.9g abi %.9g %.9g %.9g\n", pose, aff, loss, p2, a2, l2);
  }
  // (3) refine_structure (main.cpp:131-171) with gnina's quasi_newton on non_cache_cnn(HipCNNScorer), vs
  //     mi_cnn_refine_batch from the same start
  {
    minimization_params mp;
    mp.maxiters = 10;
    conf c = c0;
    c.ligands[0].rigid.position += vec(0.7, -0.4, 0.5);
    output_type out(c, 0);
    change g(m.get_size(), nc.move_receptor());
    m.set(c);
    const fl loss0 = nc.eval(m, v[1]);
    nc.adjust_center(m);
    quasi_newton qn(mp);
    const fl slope_orig = nc.getSlope();
    fl sl = 10;
    int tries = 0;
    VINA_FOR(p, 5) {
      nc.setSlope(sl);
      qn(m, prec, nc, out, g, v, user_grid);
      m.set(out.c);
      tries++;
      if (nc.within(m)) break;
      sl *= 10;
    }
    nc.setSlope(slope_orig);
    std::vector<float> x = gnina_amd::flatten(c), e(1);
    int32_t tr = 0, ev = 0;
    CK(mi_cnn_refine_batch(vina, sc, x.data(), 1, &box, (int)mp.maxiters, e.data(), &tr, &ev));
    std::printf("refine start %.9g gnina_on_seam %.9g tries %d abi %.9g tries %d evals %d\n", loss0, out.e, tries, e[0],
                (int)tr, (int)ev);
  }
  // (4) fresh_copy (main.cpp:1438, parallel_mc.cpp:146): a copy scores like the original
  {
    std::shared_ptr<DLScorer> copy = cnn.fresh_copy();
    m.set(c0);
    float a1 = 0, l1 = 0, v1 = 0, a2 = 0, l2 = 0, v2 = 0;
    const float p1 = cnn.score(m, false, a1, l1, v1), p2 = copy->score(m, false, a2, l2, v2);
    std::printf("fresh_copy %.9g %.9g | %.9g %.9g\n", p1, a1, p2, a2);
  }
  mi_vina_destroy(vina);
  mi_scorer_destroy(sc);
  mi_model_release(mm);
  return 0;
}

int main(int argc, char **argv) {
  try {
    return run(argc, argv);
  } catch (const internal_error &e) {
    fprintf(stderr, "internal_error: %s (%u)\n", e.file.c_str(), e.line);
  } catch (const usage_error &e) {
    fprintf(stderr, "usage_error: %s\n", e.what());
  } catch (const std::exception &e) {
    fprintf(stderr, "exception: %s\n", e.what());
  }
  return 3;
}
