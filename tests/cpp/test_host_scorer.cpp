// Drives the C++ host adapters (HipCNNScorer : DLScorer) the way gnina drives CNNTorchScorer:
// builds a `model` from a typed-atom file written by the pytest harness, scores every pose with
// score(model&, ...) one at a time (reference behaviour) and all at once with score_poses, and
// prints the numbers for the harness to compare with the oracle / goldens.
//
// input file (little endian): int32 n_rec, n_lig, n_poses, n_models; n_models x (int32 len, chars);
//   float rec_xyz[n_rec][3]; int32 rec_smt[n_rec]; int32 lig_smt[n_lig]; float poses[n_poses][n_lig][3]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../../gnina_amd/host/hip_cnn_scorer.h"

template <typename T> static void rd(std::ifstream &f, T *p, size_t n) { f.read((char *)p, sizeof(T) * n); }

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s atoms.bin weights_dir [--expect-invalid-name]\n", argv[0]);
    return 2;
  }
  gnina_amd::set_builtin_model_dir(argv[2]);
  if (argc > 3 && std::string(argv[3]) == "--names") {   // no GPU needed
    for (auto &n : gnina_amd::builtin_model_names()) std::printf("%s\n", n.c_str());
    return 0;
  }
  if (mi_gnina_init(0) != MI_OK) {
    std::fprintf(stderr, "init failed: %s\n", mi_last_error());
    return 3;
  }
  std::ifstream f(argv[1], std::ios::binary);
  int32_t hdr[4];
  rd(f, hdr, 4);
  const int n_rec = hdr[0], n_lig = hdr[1], n_poses = hdr[2], n_models = hdr[3];
  cnn_options opts;
  for (int i = 0; i < n_models; i++) {
    int32_t len;
    rd(f, &len, 1);
    std::string s(len, ' ');
    f.read(&s[0], len);
    opts.cnn_model_names.push_back(s);
  }
  std::vector<float> rec_xyz(3 * n_rec), poses((size_t)n_poses * n_lig * 3);
  std::vector<int32_t> rec_smt(n_rec), lig_smt(n_lig);
  rd(f, rec_xyz.data(), rec_xyz.size());
  rd(f, rec_smt.data(), rec_smt.size());
  rd(f, lig_smt.data(), lig_smt.size());
  rd(f, poses.data(), poses.size());

  try {
    cnn_options bad = opts;
    bad.cnn_model_names = {"no_such_model"};
    gnina_amd::HipCNNScorer nope(bad);
    std::printf("ERROR: invalid model name accepted\n");
    return 1;
  } catch (const usage_error &e) {
    std::printf("usage_error_ok %s\n", e.what());
  }

  if (argc > 4 && std::string(argv[3]) == "--flex") {
    // flexible residues: the first K receptor atoms are movable atoms in front of the ligand
    // (model.h: movable = [flex residues ..., ligand ...]); they are displaced by +0.25 A in x before
    // scoring, and score(compute_gradient = true) must return forces on them as well
    const int K = std::atoi(argv[4]);
    gnina_amd::HipCNNScorer scorer(opts);
    model m;
    for (int i = 0; i < K + n_lig; i++) {
      atom a;
      a.sm = i < K ? rec_smt[i] : lig_smt[i - K];
      m.atoms.push_back(a);
      m.coords.push_back(i < K ? vec(rec_xyz[3 * i], rec_xyz[3 * i + 1], rec_xyz[3 * i + 2])
                               : vec(poses[(size_t)(i - K) * 3], poses[(size_t)(i - K) * 3 + 1], poses[(size_t)(i - K) * 3 + 2]));
    }
    for (int i = K; i < n_rec; i++) {
      atom a;
      a.sm = rec_smt[i];
      a.coords = vec(rec_xyz[3 * i], rec_xyz[3 * i + 1], rec_xyz[3 * i + 2]);
      m.grid_atoms.push_back(a);
    }
    m.m_num_movable_atoms = K + n_lig;
    m.ligands.resize(1);
    m.ligands[0].node.begin = K;
    m.ligands[0].node.end = K + n_lig;
    float aff, loss, var;
    float s0 = scorer.score(m, false, aff, loss, var);
    std::printf("flex_rest %.9g %.9g\n", s0, aff);
    for (int i = 0; i < K; i++) m.coords[i][0] += 0.25f;
    float s1 = scorer.score(m, true, aff, loss, var);
    std::printf("flex_moved %.9g %.9g %.9g\n", s1, aff, loss);
    for (int i = 0; i < K + n_lig; i++)
      std::printf("force %d %.9g %.9g %.9g\n", i, m.minus_forces[i][0], m.minus_forces[i][1], m.minus_forces[i][2]);
    return 0;
  }

  if (argc > 3 && std::string(argv[3]) == "--torchmodel") {
    // the inner seam (torch_model.h:32-46): forward with compute_gradient, the gradient accessors, rotate
    const std::string name = opts.cnn_model_names.at(0);
    gnina_amd::HipTorchModel tm(std::string(argv[2]) + "/" + name + ".mgw", name);
    std::vector<float3> rc(n_rec), lc(n_lig);
    std::vector<smt> rt(rec_smt.begin(), rec_smt.end()), ltypes(lig_smt.begin(), lig_smt.end());
    for (int i = 0; i < n_rec; i++) rc[i] = float3{rec_xyz[3 * i], rec_xyz[3 * i + 1], rec_xyz[3 * i + 2]};
    for (int i = 0; i < n_lig; i++) lc[i] = float3{poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]};
    const vec nan_center(NAN, NAN, NAN);
    std::vector<float> o = tm.forward(rc, rt, lc, ltypes, nan_center, false, false);
    std::printf("tm_forward %.9g %.9g %.9g\n", o[0], o[1], o[2]);
    o = tm.forward(rc, rt, lc, ltypes, nan_center, false, true);
    std::vector<gfloat3> gl, gr;
    tm.getLigandGradient(gl);
    tm.getReceptorGradient(gr);
    std::printf("tm_grad %.9g %.9g %.9g n_lig %zu n_rec %zu\n", o[0], o[1], o[2], gl.size(), gr.size());
    for (size_t i = 0; i < gl.size(); i++) std::printf("gl %zu %.9g %.9g %.9g\n", i, gl[i].x, gl[i].y, gl[i].z);
    int shown = 0;
    for (size_t i = 0; i < gr.size() && shown < 40; i++)
      if (gr[i].x != 0 || gr[i].y != 0 || gr[i].z != 0) {
        std::printf("gr %zu %.9g %.9g %.9g\n", i, gr[i].x, gr[i].y, gr[i].z);
        shown++;
      }
    tm.seed_rotations(5);
    o = tm.forward(rc, rt, lc, ltypes, nan_center, true, false);
    gnina_amd::RotationStream rs;
    rs.seed(5);
    float q[4];
    rs.next(q);
    std::printf("tm_rotated %.9g %.9g %.9g quat %.9g %.9g %.9g %.9g\n", o[0], o[1], o[2], q[0], q[1], q[2], q[3]);
    return 0;
  }

  if (argc > 5 && std::string(argv[3]) == "--rotations") {
    // --cnn_rotation N with --seed S (cnn_torch_scorer.cpp:117-193): averaged score / affinity / variance / forces
    opts.cnn_rotations = (unsigned)std::atoi(argv[4]);
    opts.seed = (unsigned)std::atoi(argv[5]);
    gnina_amd::HipCNNScorer scorer(opts);
    model m;
    for (int i = 0; i < n_rec; i++) {
      atom a;
      a.sm = rec_smt[i];
      a.coords = vec(rec_xyz[3 * i], rec_xyz[3 * i + 1], rec_xyz[3 * i + 2]);
      m.grid_atoms.push_back(a);
    }
    for (int i = 0; i < n_lig; i++) {
      atom a;
      a.sm = lig_smt[i];
      m.atoms.push_back(a);
      m.coords.push_back(vec(poses[(size_t)i * 3], poses[(size_t)i * 3 + 1], poses[(size_t)i * 3 + 2]));
    }
    m.m_num_movable_atoms = n_lig;
    m.ligands.resize(1);
    m.ligands[0].node.begin = 0;
    m.ligands[0].node.end = n_lig;
    gnina_amd::RotationStream rs;
    rs.seed(opts.seed);
    std::printf("quat 0 1 0 0 0\n");
    for (unsigned r = 1; r < opts.cnn_rotations; r++) {
      float q[4];
      rs.next(q);
      std::printf("quat %u %.9g %.9g %.9g %.9g\n", r, q[0], q[1], q[2], q[3]);
    }
    float aff, loss, var;
    float s0 = scorer.score(m, true, aff, loss, var);
    std::printf("rot_score %.9g %.9g %.9g %.9g\n", s0, aff, loss, var);
    for (int i = 0; i < n_lig; i++)
      std::printf("force %d %.9g %.9g %.9g\n", i, m.minus_forces[i][0], m.minus_forces[i][1], m.minus_forces[i][2]);
    float s1 = scorer.score(m, false, aff, loss, var);   // same seed -> same orientations -> same numbers
    std::printf("rot_again %.9g %.9g %.9g\n", s1, aff, var);
    return 0;
  }

  if (argc > 3 && std::string(argv[3]) == "--cov") {
    // covalent docking: no ligand in the model, the atoms flagged `iscov` are what the CNN sees as the ligand
    gnina_amd::HipCNNScorer scorer(opts);
    model m;
    for (int i = 0; i < n_rec; i++) {
      atom a;
      a.sm = rec_smt[i];
      a.coords = vec(rec_xyz[3 * i], rec_xyz[3 * i + 1], rec_xyz[3 * i + 2]);
      m.grid_atoms.push_back(a);
    }
    for (int i = 0; i < n_lig; i++) {
      atom a;
      a.sm = lig_smt[i];
      a.iscov = true;
      m.atoms.push_back(a);
      m.coords.push_back(vec(poses[(size_t)i * 3], poses[(size_t)i * 3 + 1], poses[(size_t)i * 3 + 2]));
    }
    m.m_num_movable_atoms = n_lig;
    float aff, loss, var;
    float s0 = scorer.score(m, false, aff, loss, var);
    std::printf("cov %.9g %.9g %.9g\n", s0, aff, loss);
    return 0;
  }

  gnina_amd::HipCNNScorer scorer(opts);
  // model: rigid receptor in grid_atoms, ligand as the only movable atoms (ligands[0].node.begin = 0)
  model m;
  for (int i = 0; i < n_rec; i++) {
    atom a;
    a.sm = rec_smt[i];
    a.coords = vec(rec_xyz[3 * i], rec_xyz[3 * i + 1], rec_xyz[3 * i + 2]);
    m.grid_atoms.push_back(a);
  }
  for (int i = 0; i < n_lig; i++) {
    atom a;
    a.sm = lig_smt[i];
    m.atoms.push_back(a);
    m.coords.push_back(vec());
  }
  m.m_num_movable_atoms = n_lig;
  m.ligands.resize(1);
  m.ligands[0].node.begin = 0;
  m.ligands[0].node.end = n_lig;

  std::printf("models %zu initialized %d has_affinity %d\n", scorer.num_models(), (int)scorer.initialized(),
              (int)scorer.has_affinity());
  auto copy = scorer.fresh_copy();
  for (int b = 0; b < n_poses; b++) {
    for (int i = 0; i < n_lig; i++)
      m.coords[i] = vec(poses[((size_t)b * n_lig + i) * 3], poses[((size_t)b * n_lig + i) * 3 + 1],
                        poses[((size_t)b * n_lig + i) * 3 + 2]);
    float aff, loss, var;
    float s = scorer.score(m, false, aff, loss, var);
    float var2;
    float s2 = copy->score(m, var2);   // a fresh copy must give identical numbers
    std::printf("single %d %.9g %.9g %.9g %.9g copy %.9g\n", b, s, aff, loss, var, s2);
  }
  {  // compute_gradient = true (refinement): minus_forces receive d loss / d x of the heavy ligand atoms
    for (int i = 0; i < n_lig; i++)
      m.coords[i] = vec(poses[(size_t)i * 3], poses[(size_t)i * 3 + 1], poses[(size_t)i * 3 + 2]);
    float aff, loss, var;
    try {
      float s = scorer.score(m, true, aff, loss, var);
      double nrm = 0;
      for (auto &f : m.minus_forces) nrm += f[0] * f[0] + f[1] * f[1] + f[2] * f[2];
      std::printf("grad %.9g %.9g %.9g\n", s, loss, std::sqrt(nrm));
    } catch (const internal_error &e) {
      std::printf("grad_unsupported %s\n", e.what());
    }
  }
  std::vector<float> p, a, l, v;
  scorer.score_poses(m, poses, n_poses, p, a, l, v);
  for (int b = 0; b < n_poses; b++) std::printf("batch %d %.9g %.9g %.9g %.9g\n", b, p[b], a[b], l[b], v[b]);
  scorer.set_center_from_model(m);
  grid_dims box;
  scorer.set_bounding_box(box);
  std::printf("box %.6g %.6g %zu center %.6g %.6g %.6g\n", box[0].begin, box[0].end, box[0].n,
              scorer.get_center()[0], scorer.get_center()[1], scorer.get_center()[2]);
  return 0;
}
