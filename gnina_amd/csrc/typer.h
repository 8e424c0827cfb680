// typer.h -- smina type table + FileMappedGninaTyper equivalent + per-type density constants.
//
// Reference: smina type enum / names / xs_radius gninasrc/lib/atom_constants.h:45-75,101-133;
// map semantics = libmolgrid FileMappedGninaTyper as used at gninasrc/lib/torch_model.cpp:110-142.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mig {

constexpr int kNumSminaTypes = 28;

const char *smina_type_name(int smt);
float smina_xs_radius(int smt);

struct TypeMap {
  int32_t chan_of_smt[kNumSminaTypes];
  int n_channels = 0;
  // lines: one vector of type names per channel
  void build(const std::vector<std::vector<std::string>> &lines);
  bool operator==(const TypeMap &o) const;
};

// Per smina type constants of the density function (GridMaker::calc_point restated, SURVEY
// App. A.2): everything the kernel needs so that in/out decisions are bit-identical to a
// sqrtf-based CPU evaluation while the kernel itself never takes a correctly-rounded sqrt.
struct DensityConsts {
  float ar;     // radius * radius_scale
  float t2;     // smallest float x with sqrtf(x) >= ar*1.5f   (density == 0  <=>  rsq >= t2)
  float g2;     // largest  float x with sqrtf(x) <= ar        (gaussian     <=>  rsq <= g2)
  float kexp;   // -2 / (ar*ar) * log2(e): density = exp2(rsq * kexp) in the gaussian zone
  float maxr;   // ar * 1.5f
  float inv_ar; // 1 / ar
};
DensityConsts density_consts(float radius, float radius_scale);

}  // namespace mig
