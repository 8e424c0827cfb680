// model.h -- MIGNINA1 blob parsing and the layer program of one network.
//
// Replaces what TorchModel's constructor does with torch::jit::load + JSON metadata
// (gninasrc/lib/torch_model.cpp:49-118).  Blob format: gnina_amd/tools/extract_weights.py.
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "common.h"
#include "typer.h"

namespace mig {

enum class OpKind { Pool, Conv, GMax, Fc, Overlap };

struct BufDecl {
  int S = 0;  // spatial points per side
  int C = 0;  // channel capacity
};

struct Op {
  OpKind kind;
  // Pool: mode 1 = max, 2 = avg
  int pool_mode = 0;
  int src = -1, dst = -1;
  // Conv
  int ksize = 0, cin = 0, cout = 0, dst_c0 = 0, relu = 0;
  long w_off = -1, b_off = -1, bn_scale_off = -1, bn_shift_off = -1;
  // Fc
  int n_in = 0;
};

struct ModelDesc {
  std::string name, family;
  float resolution = 0.5f, dimension = 23.5f, radius_scaling = 1.0f;
  bool skip_softmax = false, apply_logistic_loss = false;
  TypeMap recmap, ligmap;
  std::vector<BufDecl> bufs;
  std::vector<Op> ops;
  std::vector<float> data;  // raw fp32 payload
  int grid_points() const;
  int n_channels() const { return recmap.n_channels + ligmap.n_channels; }
};

// Throws mig::Error(MI_ERR_MODEL) on malformed input.
ModelDesc parse_blob(const void *blob, size_t nbytes, const char *name_override);

// Re-target a parsed model to another grid resolution / dimension (dynamic-pool families only).
void regrid(ModelDesc &m, float resolution, float dimension);

}  // namespace mig
