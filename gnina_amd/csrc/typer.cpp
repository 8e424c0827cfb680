#include "typer.h"

#include <cmath>
#include <cstring>

namespace mig {

// atom_constants.h:101-133, columns smina_name and xs_radius (data table, restated).
static const char *const kNames[kNumSminaTypes] = {
    "Hydrogen", "PolarHydrogen", "AliphaticCarbonXSHydrophobe", "AliphaticCarbonXSNonHydrophobe",
    "AromaticCarbonXSHydrophobe", "AromaticCarbonXSNonHydrophobe", "Nitrogen", "NitrogenXSDonor",
    "NitrogenXSDonorAcceptor", "NitrogenXSAcceptor", "Oxygen", "OxygenXSDonor", "OxygenXSDonorAcceptor",
    "OxygenXSAcceptor", "Sulfur", "SulfurAcceptor", "Phosphorus", "Fluorine", "Chlorine", "Bromine", "Iodine",
    "Magnesium", "Manganese", "Zinc", "Calcium", "Iron", "GenericMetal", "Boron"};

static const float kXsRadius[kNumSminaTypes] = {0.37f, 0.37f, 1.9f, 1.9f, 1.9f, 1.9f, 1.8f, 1.8f, 1.8f, 1.8f,
                                                1.7f,  1.7f,  1.7f, 1.7f, 2.0f, 2.0f, 2.1f, 1.5f, 1.8f, 2.0f,
                                                2.2f,  1.2f,  1.2f, 1.2f, 1.2f, 1.2f, 1.2f, 1.92f};

const char *smina_type_name(int smt) { return (smt >= 0 && smt < kNumSminaTypes) ? kNames[smt] : nullptr; }
float smina_xs_radius(int smt) { return (smt >= 0 && smt < kNumSminaTypes) ? kXsRadius[smt] : 0.0f; }

void TypeMap::build(const std::vector<std::vector<std::string>> &lines) {
  for (int i = 0; i < kNumSminaTypes; i++) chan_of_smt[i] = -1;
  n_channels = 0;
  for (const auto &line : lines) {
    if (line.empty()) continue;
    for (const auto &nm : line) {
      int hit = -1;
      for (int t = 0; t < kNumSminaTypes; t++)
        if (nm == kNames[t]) hit = t;
      if (hit < 0) throw std::string("unknown smina type name in map: " + nm);
      chan_of_smt[hit] = n_channels;
    }
    n_channels++;
  }
}

bool TypeMap::operator==(const TypeMap &o) const {
  return n_channels == o.n_channels && std::memcmp(chan_of_smt, o.chan_of_smt, sizeof(chan_of_smt)) == 0;
}

DensityConsts density_consts(float radius, float radius_scale) {
  DensityConsts k;
  volatile float ar = radius * radius_scale;  // volatile: keep every step a rounded fp32 op
  k.ar = ar;
  volatile float maxr = ar * 1.5f;
  k.maxr = maxr;
  // smallest x with sqrtf(x) >= maxr
  float x = maxr * maxr;
  while (sqrtf(x) >= maxr) x = std::nextafterf(x, -INFINITY);
  while (sqrtf(x) < maxr) x = std::nextafterf(x, INFINITY);
  k.t2 = x;
  // largest x with sqrtf(x) <= ar
  float y = ar * ar;
  while (sqrtf(y) <= ar) y = std::nextafterf(y, INFINITY);
  while (sqrtf(y) > ar) y = std::nextafterf(y, -INFINITY);
  k.g2 = y;
  k.kexp = (float)(-2.0 / ((double)ar * (double)ar) * 1.4426950408889634);
  k.inv_ar = 1.0f / ar;
  return k;
}

}  // namespace mig
