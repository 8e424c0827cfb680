#include "hip_cnn_scorer.h"

#include <dirent.h>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace gnina_amd {

static std::string g_model_dir = "gnina_amd/weights";
void set_builtin_model_dir(const std::string &dir) { g_model_dir = dir; }
std::string builtin_model_dir() { return g_model_dir; }

std::vector<std::string> builtin_model_names() {
  std::vector<std::string> out;
  if (DIR *d = opendir(g_model_dir.c_str())) {
    while (dirent *e = readdir(d)) {
      std::string f = e->d_name;
      if (f.size() > 4 && f.substr(f.size() - 4) == ".mgw") out.push_back(f.substr(0, f.size() - 4));
    }
    closedir(d);
  }
  std::sort(out.begin(), out.end());
  return out;
}

// ---------------------------------------------------------------------------------------------
HipTorchModel::HipTorchModel(const std::string &path, const std::string &name) {
  model_ = mi_model_load_file(path.c_str());
  if (!model_) throw usage_error("Could not read torch model " + name + ": " + mi_last_error());  // torch_model.cpp:115-117
  mi_model_info(model_, &res_, &dim_, nullptr, nullptr, nullptr);
  scorer_ = mi_scorer_create(&model_, 1);
  if (!scorer_) throw internal_error(mi_last_error(), 0);
}

HipTorchModel::~HipTorchModel() {
  mi_scorer_destroy(scorer_);
  mi_model_release(model_);
}

void RotationStream::next(float q[4]) {
  // libmolgrid Transform: u1, u2, u3 ~ U[0, 1) in double; Q = (sqrt(1-u1) sin 2 pi u2, sqrt(1-u1) cos 2 pi u2,
  // sqrt(u1) sin 2 pi u3, sqrt(u1) cos 2 pi u3)
  std::uniform_real_distribution<double> unit_sample(0, 1);
  const double u1 = unit_sample(engine), u2 = unit_sample(engine), u3 = unit_sample(engine);
  const double sq1 = std::sqrt(1 - u1), sqr = std::sqrt(u1), two_pi = 2 * 3.14159265358979323846;
  q[0] = (float)(sq1 * std::sin(two_pi * u2));
  q[1] = (float)(sq1 * std::cos(two_pi * u2));
  q[2] = (float)(sqr * std::sin(two_pi * u3));
  q[3] = (float)(sqr * std::cos(two_pi * u3));
  const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}

std::vector<float> HipTorchModel::forward(const std::vector<float3> &rec_coords, const std::vector<smt> &rec_types,
                                          const std::vector<float3> &lig_coords, const std::vector<smt> &lig_types,
                                          const vec &center, bool rotate, bool compute_gradient) {
  if (rec_coords.size() != rec_types.size() || lig_coords.size() != lig_types.size())
    throw internal_error("Shape mismatch", 0);  // torch_model.cpp:122-123
  if (rec_key_ != rec_coords.data() || rec_n_ != rec_coords.size()) {  // receptor assumed constant (dl_scorer.cpp:112)
    std::vector<int32_t> t(rec_types.begin(), rec_types.end());
    if (mi_scorer_set_receptor(scorer_, &rec_coords[0].x, t.data(), (int)t.size()) != MI_OK)
      throw internal_error(mi_last_error(), 0);
    rec_key_ = rec_coords.data();
    rec_n_ = rec_coords.size();
    all_rows_flex_ = false;
  }
  std::vector<int32_t> lt(lig_types.begin(), lig_types.end());
  float c[3] = {center[0], center[1], center[2]};
  const float *cptr = std::isfinite(c[0]) ? c : nullptr;  // non-finite: lig.center() (torch_model.cpp:163-166)
  if (rotate) {
    float q[4];
    rot_.next(q);
    if (mi_scorer_set_rotations(scorer_, q, 1) != MI_OK) throw internal_error(mi_last_error(), 0);
  }
  float pose, aff, loss;
  if (!compute_gradient) {
    if (mi_scorer_score_batch(scorer_, &lig_coords[0].x, lt.data(), 1, (int)lt.size(), cptr, &pose, &aff, &loss,
                              nullptr) != MI_OK)
      throw internal_error(mi_last_error(), 0);
    return {pose, aff, loss};
  }
  // gradient for every receptor atom, like the reference's coord_grad (torch_model.cpp:201-219): all rows movable
  const int nr = (int)rec_coords.size(), nl = (int)lig_coords.size();
  if (!all_rows_flex_) {
    std::vector<int32_t> rows(nr);
    for (int i = 0; i < nr; i++) rows[i] = i;
    if (mi_scorer_set_flex(scorer_, rows.data(), nr) != MI_OK) throw internal_error(mi_last_error(), 0);
    all_rows_flex_ = true;
  }
  gradient_lig.assign(nl, gfloat3{0, 0, 0});
  gradient_rec.assign(nr, gfloat3{0, 0, 0});
  float var;
  if (mi_scorer_score_flex(scorer_, &lig_coords[0].x, lt.data(), 1, nl, cptr, nr ? &rec_coords[0].x : nullptr, &pose, &aff,
                           &loss, &var, nl ? &gradient_lig[0].x : nullptr, nr ? &gradient_rec[0].x : nullptr) != MI_OK)
    throw internal_error(mi_last_error(), 0);
  return {pose, aff, loss};
}

void HipTorchModel::getLigandGradient(std::vector<gfloat3> &grad) { grad = gradient_lig; }    // torch_model.cpp:226-231
void HipTorchModel::getReceptorGradient(std::vector<gfloat3> &grad) { grad = gradient_rec; }

// ---------------------------------------------------------------------------------------------
// DLScorer pieces (stand-alone build only; inside gnina the real dl_scorer.cpp provides them)
#ifndef MI_GNINA_WITH_GNINA_HEADERS
}  // namespace gnina_amd

// Ligand = movable atoms from the first ligand's root to m_num_movable_atoms, hydrogens included
// (dl_scorer.cpp:71-87); without a ligand (covalent docking, :43-69) the atoms flagged `iscov` among the
// flexible-residue and inflex atoms play the ligand's role.
void DLScorer::setLigand(const model &m) {
  num_atoms = m.atoms.size();
  if (m.ligands.empty() && m.coords.empty()) return;
  if (m.ligands.empty()) {
    ligand_smtypes.clear();
    ligand_coords.clear();
    ligand_map.clear();
    auto take = [&](sz i) {
      if (!m.atoms[i].iscov) return;
      const vec &c = m.coords[i];
      ligand_smtypes.push_back(m.atoms[i].sm);
      ligand_coords.push_back(float3{c[0], c[1], c[2]});
      ligand_map.push_back((int)i);
    };
    for (sz i = 0; i < m.m_num_movable_atoms; i++) take(i);
    for (sz i = m.m_num_movable_atoms; i < m.coords.size(); i++) take(i);
    return;
  }
  const sz off = m.ligands[0].node.begin;
  const sz n = m.m_num_movable_atoms - off;
  ligand_smtypes.resize(n);
  ligand_coords.resize(n);
  ligand_map.resize(n);
  for (sz i = 0; i < n; i++) {
    ligand_smtypes[i] = m.atoms[i + off].sm;
    const vec &c = m.coords[i + off];
    ligand_coords[i] = float3{c[0], c[1], c[2]};
    ligand_map[i] = (int)(i + off);
  }
}

// Receptor = flexible-residue movable atoms (before the ligand), then inflex atoms, then fixed atoms, covalent
// (`iscov`) atoms left out; types cached on the first call, flexible coordinates refreshed afterwards
// (dl_scorer.cpp:93-193).
void DLScorer::setReceptor(const model &m) {
  num_atoms = m.atoms.size();
  const sz n_flex = m.ligands.empty() ? m.m_num_movable_atoms : m.ligands[0].node.begin;
  const sz n_total = n_flex + (m.atoms.size() - m.m_num_movable_atoms) + m.grid_atoms.size();
  if (receptor_smtypes.empty()) {
    for (sz i = 0; i < n_flex; i++)
      if (!m.atoms[i].iscov) receptor_smtypes.push_back(m.atoms[i].sm);
    for (sz i = m.m_num_movable_atoms; i < m.atoms.size(); i++)
      if (!m.atoms[i].iscov) receptor_smtypes.push_back(m.atoms[i].sm);
    for (const atom &a : m.grid_atoms) receptor_smtypes.push_back(a.sm);
  }
  if (receptor_coords.empty()) {
    for (sz i = 0; i < n_flex; i++)
      if (!m.atoms[i].iscov) {
        receptor_coords.push_back(float3{m.coords[i][0], m.coords[i][1], m.coords[i][2]});
        receptor_map.push_back((int)i);
      }
    for (sz i = m.m_num_movable_atoms; i < m.coords.size(); i++)
      if (!m.atoms[i].iscov) receptor_coords.push_back(float3{m.coords[i][0], m.coords[i][1], m.coords[i][2]});
    for (const atom &a : m.grid_atoms) receptor_coords.push_back(float3{a.coords[0], a.coords[1], a.coords[2]});
  } else if (receptor_coords.size() == n_total) {  // (as in the reference: only when no atom was left out)
    for (sz i = 0; i < n_flex; i++)
      if (!m.atoms[i].iscov) receptor_coords[i] = float3{m.coords[i][0], m.coords[i][1], m.coords[i][2]};
  }
}

// Centre of the heavy movable atoms (dl_scorer.cpp:197-217)
void DLScorer::set_center_from_model(model &m) {
  current_center = vec(0, 0, 0);
  unsigned cnt = 0;
  for (const vec &c : m.get_heavy_atom_movable_coords()) {
    current_center[0] += c[0];
    current_center[1] += c[1];
    current_center[2] += c[2];
    cnt++;
  }
  for (int i = 0; i < 3; i++) current_center[i] /= (float)cnt;
}

namespace gnina_amd {
#endif

static bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// Model-name handling of CNNTorchScorer's constructor (cnn_torch_scorer.cpp:24-92): default
// ensemble, "fast", "default1.0", "<prefix>_ensemble" expansion, external files.
HipCNNScorer::HipCNNScorer(const cnn_options &opts) : DLScorer(opts) {
  if (cnnopts.cnn_scoring == CNNnone) return;
  if (cnnopts.cnn_models.empty()) {
    auto &names = cnnopts.cnn_model_names;
    if (names.empty()) {
      names = {"dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"};
    } else if (names.size() == 1 && names[0] == "fast") {
      names[0] = "all_default_to_default_1_3_1";
    } else if (names.size() == 1 && names[0] == "default1.0") {
      names = {"dense", "general_default2018_3", "dense_3", "crossdock_default2018", "redock_default2018_2"};
    }
  }
  const std::vector<std::string> avail = builtin_model_names();
  std::vector<std::string> expanded;
  for (const std::string &name : cnnopts.cnn_model_names) {
    if (ends_with(name, "_ensemble")) {
      const std::string prefix = name.substr(0, name.size() - 9);
      for (const std::string &a : avail)
        if (a.compare(0, prefix.size(), prefix) == 0) expanded.push_back(a);
    } else {
      expanded.push_back(name);
    }
  }
  for (const std::string &name : expanded) {
    if (std::find(avail.begin(), avail.end(), name) == avail.end()) throw usage_error("Invalid model name: " + name);
    models.push_back(std::make_shared<HipTorchModel>(g_model_dir + "/" + name + ".mgw", name));
  }
  for (const std::string &f : cnnopts.cnn_models) models.push_back(std::make_shared<HipTorchModel>(f, f));
  if (!models.empty()) {
    std::vector<mi_model *> hs;
    for (auto &m : models) hs.push_back(m->handle());
    mi_scorer *s = mi_scorer_create(hs.data(), (int)hs.size());
    if (!s) throw internal_error(mi_last_error(), 0);
    ensemble.reset(s, mi_scorer_destroy);
  }
}

std::shared_ptr<DLScorer> HipCNNScorer::fresh_copy() const {
  // cheap by construction: device weights are shared and refcounted; only the per-copy
  // workspace and receptor binding are new (the reference reloads every model, main.cpp:1438)
  auto c = std::make_shared<HipCNNScorer>();
  c->cnnopts = cnnopts;
  c->models = models;
  if (!models.empty()) {
    std::vector<mi_model *> hs;
    for (auto &m : models) hs.push_back(m->handle());
    c->ensemble.reset(mi_scorer_create(hs.data(), (int)hs.size()), mi_scorer_destroy);
  }
  return c;
}

// Types and coordinates go to the device once (the receptor is assumed constant, dl_scorer.cpp:112,150);
// the flexible-residue rows -- the first receptor_map.size() rows -- are declared so that every call can
// pass their current coordinates (dl_scorer.cpp:181-192).
void HipCNNScorer::upload_receptor() {
  if (receptor_uploaded) return;
  std::vector<int32_t> t(receptor_smtypes.begin(), receptor_smtypes.end());
  if (mi_scorer_set_receptor(ensemble.get(), &receptor_coords[0].x, t.data(), (int)t.size()) != MI_OK)
    throw internal_error(mi_last_error(), 0);
  if (!receptor_map.empty()) {
    std::vector<int32_t> rows(receptor_map.size());
    for (size_t i = 0; i < rows.size(); i++) rows[i] = (int32_t)i;
    if (mi_scorer_set_flex(ensemble.get(), rows.data(), (int)rows.size()) != MI_OK)
      throw internal_error(mi_last_error(), 0);
  }
  receptor_uploaded = true;
}

float HipCNNScorer::score(model &m, float &variance) {
  float aff = 0, loss = 0;
  return score(m, false, aff, loss, variance);
}

void HipCNNScorer::score_poses(model &m, const std::vector<float> &lig_xyz, int B, std::vector<float> &pose,
                               std::vector<float> &affinity, std::vector<float> &loss,
                               std::vector<float> &variance) {
  if (!initialized()) throw internal_error("scorer not initialised", 0);
  setLigand(m);
  setReceptor(m);
  upload_receptor();
  const int L = (int)ligand_smtypes.size();
  if ((size_t)B * L * 3 != lig_xyz.size()) throw internal_error("Shape mismatch", 0);
  std::vector<int32_t> lt(ligand_smtypes.begin(), ligand_smtypes.end());
  std::vector<float> centers;
  const float *cptr = nullptr;
  if (!std::isnan(cnnopts.cnn_center[0])) {  // cnn_torch_scorer.cpp:137-140
    centers.resize((size_t)B * 3);
    for (int b = 0; b < B; b++)
      for (int k = 0; k < 3; k++) centers[3 * b + k] = cnnopts.cnn_center[k];
    cptr = centers.data();
  }
  pose.resize(B);
  affinity.resize(B);
  loss.resize(B);
  variance.resize(B);
  mi_status st;
  if (receptor_map.empty()) {
    st = mi_scorer_score_batch(ensemble.get(), lig_xyz.data(), lt.data(), B, L, cptr, pose.data(), affinity.data(),
                               loss.data(), variance.data());
  } else {  // every pose sees the flexible residues where the model has them now
    const size_t nf = receptor_map.size();
    std::vector<float> flex((size_t)B * nf * 3);
    for (int b = 0; b < B; b++) std::memcpy(&flex[(size_t)b * nf * 3], receptor_coords.data(), nf * 3 * sizeof(float));
    st = mi_scorer_score_flex(ensemble.get(), lig_xyz.data(), lt.data(), B, L, cptr, flex.data(), pose.data(),
                              affinity.data(), loss.data(), variance.data(), nullptr, nullptr);
  }
  if (st != MI_OK) throw internal_error(mi_last_error(), 0);
}

// CNNTorchScorer::score (cnn_torch_scorer.cpp:105-198).  The reference loops models x rotations with one
// TorchModel::forward each; here the max(cnn_rotations, 1) orientations of the pose form ONE batch through the whole
// ensemble (rotation 0 = as is, the others from the rotation stream re-seeded with cnn_options::seed -- the same
// orientations for every model, as the reference re-seeds per model), and the per-model outputs are then accumulated
// on the host in the reference's order (model-major; score in double; population variance of all affinities).
float HipCNNScorer::score(model &m, bool compute_gradient, float &affinity, float &loss, float &variance) {
  if (!initialized()) return -1.0;  // cnn_torch_scorer.cpp:107-108
  setLigand(m);
  setReceptor(m);
  upload_receptor();
  m.clear_minus_forces();  // "ALERT: clears minus forces" (cnn_torch_scorer.cpp:115)
  const int n_rot = (int)std::max(cnnopts.cnn_rotations, 1U), nm = (int)models.size();
  const int L = (int)ligand_smtypes.size();
  std::vector<int32_t> lt(ligand_smtypes.begin(), ligand_smtypes.end());
  std::vector<float> xyz((size_t)n_rot * L * 3);
  for (int r = 0; r < n_rot; r++) std::memcpy(&xyz[(size_t)r * L * 3], ligand_coords.data(), (size_t)L * 3 * sizeof(float));
  if (n_rot > 1) {
    std::vector<float> quats((size_t)n_rot * 4, 0.f);
    quats[0] = 1.f;  // r == 0: rotate = false
    RotationStream rs;
    rs.seed(cnnopts.seed);
    for (int r = 1; r < n_rot; r++) rs.next(&quats[4 * r]);
    if (mi_scorer_set_rotations(ensemble.get(), quats.data(), n_rot) != MI_OK) throw internal_error(mi_last_error(), 0);
  }
  std::vector<float> centers;
  const float *cptr = nullptr;
  if (!std::isnan(cnnopts.cnn_center[0])) {  // cnn_torch_scorer.cpp:137-140
    centers.resize((size_t)n_rot * 3);
    for (int r = 0; r < n_rot; r++)
      for (int k = 0; k < 3; k++) centers[3 * r + k] = cnnopts.cnn_center[k];
    cptr = centers.data();
  }
  const size_t nf = receptor_map.size();
  std::vector<float> flex(n_rot * nf * 3);
  for (int r = 0; r < n_rot && nf; r++) std::memcpy(&flex[(size_t)r * nf * 3], receptor_coords.data(), nf * 3 * sizeof(float));
  std::vector<float> p(n_rot), a(n_rot), l(n_rot), v(n_rot), lg, fg;
  if (compute_gradient) {
    lg.resize((size_t)n_rot * L * 3);
    fg.resize((size_t)n_rot * nf * 3);
  }
  mi_status st;
  if (nf || compute_gradient)
    st = mi_scorer_score_flex(ensemble.get(), xyz.data(), lt.data(), n_rot, L, cptr, nf ? flex.data() : nullptr, p.data(),
                              a.data(), l.data(), v.data(), compute_gradient ? lg.data() : nullptr,
                              compute_gradient && nf ? fg.data() : nullptr);
  else
    st = mi_scorer_score_batch(ensemble.get(), xyz.data(), lt.data(), n_rot, L, cptr, p.data(), a.data(), l.data(), v.data());
  if (st != MI_OK) throw internal_error(mi_last_error(), 0);

  double score = 0.0;
  affinity = 0.0;
  loss = 0.0;
  unsigned cnt = 0;
  const unsigned nscores = (unsigned)(nm * n_rot);
  std::vector<float> affinities, pm(n_rot), am(n_rot), lm(n_rot);
  for (int mi = 0; mi < nm; mi++) {
    if (mi_scorer_last_model_outputs(ensemble.get(), mi, pm.data(), am.data(), lm.data(), n_rot) != MI_OK)
      throw internal_error(mi_last_error(), 0);
    for (int r = 0; r < n_rot; r++) {
      score += pm[r];
      if (nscores > 1) affinities.push_back(am[r]);
      affinity += am[r];
      loss += lm[r];
      cnt++;
    }
  }
  if (compute_gradient) {
    // getGradient + add_minus_forces per evaluation, scale_minus_forces(1 / cnt) at the end
    // (cnn_torch_scorer.cpp:160-176,208-228) = the mean over models (done by the engine) and rotations
    std::vector<gfloat3> gradient(receptor_map.size() + ligand_map.size(), gfloat3{0, 0, 0});
    if (gradient.size() < (size_t)m.m_num_movable_atoms) gradient.resize(m.m_num_movable_atoms, gfloat3{0, 0, 0});
    const float w = 1.0f / (float)n_rot;
    for (int r = 0; r < n_rot; r++) {
      for (sz i = 0; i < ligand_map.size(); i++) {
        gfloat3 &g = gradient[ligand_map[i]];
        const float *s3 = &lg[((size_t)r * L + i) * 3];
        g.x += w * s3[0], g.y += w * s3[1], g.z += w * s3[2];
      }
      for (sz i = 0; i < nf; i++) {
        gfloat3 &g = gradient[receptor_map[i]];
        const float *s3 = &fg[((size_t)r * nf + i) * 3];
        g.x += w * s3[0], g.y += w * s3[1], g.z += w * s3[2];
      }
    }
    m.add_minus_forces(gradient);
  }
  affinity /= cnt;
  loss /= cnt;
  score /= cnt;  // mean
  variance = 0;
  if (affinities.size() > 1) {
    float sum = 0;
    for (float s1 : affinities) {
      float diff = affinity - s1;
      diff *= diff;
      sum += diff;
    }
    variance = sum / affinities.size();
  }
  return (float)score;
}

void HipCNNScorer::set_bounding_box(grid_dims &box) const {  // cnn_torch_scorer.cpp:230-242
  vec center = get_center();
  fl dim = get_grid_dim();
  fl n = dim / get_grid_res();
  fl half = dim / 2.0;
  for (unsigned i = 0; i < 3; i++) {
    box[i].begin = center[i] - half;
    box[i].end = center[i] + half;
    box[i].n = (sz)n;
  }
}

fl HipCNNScorer::get_grid_dim() const {
  if (models.empty()) throw internal_error("no models", 0);
  return models[0]->get_grid_dim();
}
fl HipCNNScorer::get_grid_res() const {
  if (models.empty()) throw internal_error("no models", 0);
  return models[0]->get_grid_res();
}

}  // namespace gnina_amd
