// hip_cnn_scorer.h -- C++ host adapters above the C ABI, mirroring gnina's own classes.
//
//   HipTorchModel  <->  TorchModel<isCUDA>      gninasrc/lib/torch_model.h:22-47
//   HipCNNScorer   <->  CNNTorchScorer<isCUDA>  gninasrc/lib/cnn_torch_scorer.h (derives from DLScorer)
//
// Same names, argument meaning and error behaviour (usage_error for unreadable / unknown models,
// -1.0 when not initialised); the arithmetic lives in libmi_gnina.so (include/mi_gnina.h).
#pragma once
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../../include/mi_gnina.h"
#include "gnina_types.h"

namespace gnina_amd {

// directory holding <name>.mgw blobs for gnina's built-in model names (torch_models.h:19-20 analogue)
void set_builtin_model_dir(const std::string &dir);
std::string builtin_model_dir();
// names available as built-ins (file stems with '.' -> '_', make_model_cpp.py:27-29)
std::vector<std::string> builtin_model_names();

// libmolgrid's Transform(center, 0, random_rotate = true) (called at torch_model.cpp:170): a uniformly random unit
// quaternion by Shoemake's method from libmolgrid::random_engine, a std::default_random_engine that
// CNNTorchScorer::score re-seeds with cnn_options::seed for every model (cnn_torch_scorer.cpp:126-127).  libmolgrid is
// not part of the reference tree (fetched at build time): the formula below is restated from its published
// source and the stream is "unpinned"; the averaging arithmetic around it follows cnn_torch_scorer.cpp:117-193 exactly.
struct RotationStream {
  std::default_random_engine engine;
  void seed(unsigned s) { engine.seed(s); }
  void next(float q[4]);   // (a, b, c, d) of a + bi + cj + dk
};

class HipTorchModel {
  mi_model *model_ = nullptr;
  mi_scorer *scorer_ = nullptr;          // single-model scorer used by forward()
  const void *rec_key_ = nullptr;        // receptor identity cache (data pointer + size)
  size_t rec_n_ = 0;
  bool all_rows_flex_ = false;           // receptor gradient requested: every receptor row is declared movable
  float res_ = 0.5f, dim_ = 23.5f;
  std::vector<gfloat3> gradient_rec, gradient_lig;   // torch_model.h:26
  RotationStream rot_;

 public:
  HipTorchModel(const std::string &path, const std::string &name);
  ~HipTorchModel();
  HipTorchModel(const HipTorchModel &) = delete;
  HipTorchModel &operator=(const HipTorchModel &) = delete;

  // TorchModel::forward (torch_model.h:34-36, torch_model.cpp:153-224): returns {pose, affinity, loss}.
  // rotate: the atoms are turned about the grid centre by the next quaternion of the rotation stream (seed_rotations)
  // before voxelization.  compute_gradient: d loss / d coordinates of every ligand and receptor atom is kept for
  // getLigandGradient / getReceptorGradient (in the unrotated frame).
  std::vector<float> forward(const std::vector<float3> &rec_coords, const std::vector<smt> &rec_types,
                             const std::vector<float3> &lig_coords, const std::vector<smt> &lig_types,
                             const vec &center, bool rotate, bool compute_gradient);
  // "assumes forward was called with compute_gradient" (torch_model.h:38-40)
  void getLigandGradient(std::vector<gfloat3> &grad);
  void getReceptorGradient(std::vector<gfloat3> &grad);
  void seed_rotations(unsigned seed) { rot_.seed(seed); }   // libmolgrid::random_engine.seed (cnn_torch_scorer.cpp:127)
  float get_grid_dim() const { return dim_; }
  float get_grid_res() const { return res_; }
  mi_model *handle() const { return model_; }
};

class HipCNNScorer : public DLScorer {
  std::vector<std::shared_ptr<HipTorchModel>> models;
  std::shared_ptr<mi_scorer> ensemble;   // all models behind one batched scorer
  bool receptor_uploaded = false;
  void upload_receptor();

 public:
  HipCNNScorer() {}
  HipCNNScorer(const cnn_options &opts);
  virtual ~HipCNNScorer() {}

  bool initialized() const override { return !models.empty(); }
  bool has_affinity() const override { return true; }
  float score(model &m, float &variance) override;
  float score(model &m, bool compute_gradient, float &affinity, float &loss, float &variance) override;
  void set_bounding_box(grid_dims &box) const override;
  std::shared_ptr<DLScorer> fresh_copy() const override;

  // MI355X extension: score B poses of the current ligand in one call (coordinates [B][L][3] in
  // the ligand atom order of setLigand).  The reference has no batched entry point.
  void score_poses(model &m, const std::vector<float> &lig_xyz, int B, std::vector<float> &pose,
                   std::vector<float> &affinity, std::vector<float> &loss, std::vector<float> &variance);

  fl get_grid_dim() const;
  fl get_grid_res() const;
  size_t num_models() const { return models.size(); }
};

}  // namespace gnina_amd
