// oracle/_ref link stubs -- TEST INFRASTRUCTURE ONLY.  Definitions for the symbols of the reference's GPU-only
// translation units (tree_gpu.cu, conf_gpu.cu, gpucode.cu, device_buffer.cpp, non_cache_gpu.*, cache_gpu.*) that the
// CPU files reference but oracle/_ref never executes.  Every one aborts if reached.
#include <cstdio>
#include <cstdlib>
#define REF_UNREACHABLE() (std::fprintf(stderr, "oracle/_ref: GPU-only reference code reached (%s)\n", __func__), std::abort())
#include "bfgs.h"
#include "cache_gpu.h"
#include "conf_gpu.h"
#include "gpucode.h"
#include "non_cache_gpu.h"

cudaError definitelyPinnedMemcpy(void *, const void *, size_t, cudaMemcpyKind) { REF_UNREACHABLE(); }
change_gpu::change_gpu(const change &, const gpu_data &, device_buffer &) { REF_UNREACHABLE(); }
conf_gpu::conf_gpu(const conf &, const gpu_data &, device_buffer &) { REF_UNREACHABLE(); }
void conf_gpu::set_cpu(conf &, const gpu_data &) const { REF_UNREACHABLE(); }
float single_point_calc(const GPUNonCacheInfo &, atom_params *, force_energy_tup *, float) { REF_UNREACHABLE(); }
template <typename infoT>
fl bfgs(quasi_newton_aux_gpu<infoT> &, conf_gpu &, change_gpu &, const fl, const minimization_params &) {
  REF_UNREACHABLE();
}
template fl bfgs(quasi_newton_aux_gpu<GPUNonCacheInfo> &, conf_gpu &, change_gpu &, const fl, const minimization_params &);
template fl bfgs(quasi_newton_aux_gpu<GPUCacheInfo> &, conf_gpu &, change_gpu &, const fl, const minimization_params &);
non_cache_gpu::non_cache_gpu(szv_grid_cache &gcache, const grid_dims &gd_, const precalculate_gpu *, fl slope_)
    : non_cache(gcache, gd_, nullptr, slope_) {
  REF_UNREACHABLE();
}
non_cache_gpu::~non_cache_gpu() {}
fl non_cache_gpu::eval(model &, fl) const { REF_UNREACHABLE(); }
void non_cache_gpu::setSlope(fl) { REF_UNREACHABLE(); }
