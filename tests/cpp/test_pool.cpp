// mi_pool (include/mi_gnina.h): the C++ multi-device host path, driven from C++ only -- no Python, no torch.
//
//   test_pool <weights_dir> [n_devices (0 = all visible)] [B]
//
// Builds a seeded synthetic complex (the generator of SURVEY 8d config C2: receptor atoms uniform in a cube with a
// pocket, a Gaussian-blob ligand, rigid random poses), scores the batch (a) with one mi_scorer on device 0, (b) through
// an mi_pool over n devices with host buffers, (c) through the pool with everything resident on devices[0] (the RCCL
// scatter / gather path when n > 1), and prints whether (b) and (c) equal (a) bit for bit, plus poses/s of (b) and (c)
// for pool sizes 1 .. n (strong scaling of this one batch).  tests/test_host_adapter.py parses the output.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/mi_gnina.h"

#define CK(call)                                                       \
  do {                                                                 \
    if ((call) != MI_OK) {                                             \
      std::fprintf(stderr, "%s failed: %s\n", #call, mi_last_error()); \
      return 4;                                                        \
    }                                                                  \
  } while (0)
#define HK(call)                                                                    \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));        \
      return 5;                                                                     \
    }                                                                               \
  } while (0)

static bool same_bits(const std::vector<float> &a, const std::vector<float> &b) {
  return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s weights_dir [n_devices] [B]\n", argv[0]);
    return 2;
  }
  const std::string wdir = argv[1];
  int want = argc > 2 ? std::atoi(argv[2]) : 0;
  const int B = argc > 3 ? std::atoi(argv[3]) : 1024;
  const int have = mi_gnina_device_count();
  if (have <= 0) {
    std::fprintf(stderr, "no HIP device\n");
    return 3;
  }
  if (want <= 0 || want > have) want = have;
  CK(mi_gnina_init(0));

  // seeded complex: types from default2017's maps (carbon / nitrogen / oxygen flavours that every map knows)
  std::mt19937 gen(0);
  std::uniform_real_distribution<float> U(-20.f, 20.f), T(-2.f, 2.f);
  std::normal_distribution<float> N(0.f, 2.5f), N1(0.f, 1.f);
  const int rec_types[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13}, lig_types[] = {2, 3, 4, 5, 6, 8, 9, 10, 12};
  const int n_rec = 2500, L = 32;
  std::vector<float> rec_xyz;
  std::vector<int32_t> rec_smt, lig_smt(L);
  while ((int)rec_smt.size() < n_rec) {
    const float x = U(gen), y = U(gen), z = U(gen);
    if (x * x + y * y + z * z < 16.f) continue;  // the pocket
    rec_xyz.push_back(x), rec_xyz.push_back(y), rec_xyz.push_back(z);
    rec_smt.push_back(rec_types[gen() % 12]);
  }
  std::vector<float> lig0(3 * L);
  for (int i = 0; i < L; i++) {
    for (int k = 0; k < 3; k++) lig0[3 * i + k] = N(gen);
    lig_smt[i] = lig_types[gen() % 9];
  }
  std::vector<float> poses((size_t)B * L * 3);
  for (int b = 0; b < B; b++) {
    float q[4] = {N1(gen), N1(gen), N1(gen), N1(gen)};
    const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (float &v : q) v /= n;
    const float a = q[0], bb = q[1], c = q[2], d = q[3];
    const float R[9] = {a * a + bb * bb - c * c - d * d, 2 * (bb * c - a * d), 2 * (bb * d + a * c),
                        2 * (bb * c + a * d), a * a - bb * bb + c * c - d * d, 2 * (c * d - a * bb),
                        2 * (bb * d - a * c), 2 * (c * d + a * bb), a * a - bb * bb - c * c + d * d};
    const float t[3] = {T(gen), T(gen), T(gen)};
    for (int i = 0; i < L; i++)
      for (int k = 0; k < 3; k++)
        poses[((size_t)b * L + i) * 3 + k] =
            R[3 * k] * lig0[3 * i] + R[3 * k + 1] * lig0[3 * i + 1] + R[3 * k + 2] * lig0[3 * i + 2] + t[k];
  }

  const std::string path = wdir + "/default2017.mgw";
  const char *paths[1] = {path.c_str()};
  // (a) one scorer
  mi_model *m = mi_model_load_file(path.c_str());
  if (!m) {
    std::fprintf(stderr, "mi_model_load_file: %s\n", mi_last_error());
    return 4;
  }
  mi_scorer *sc = mi_scorer_create(&m, 1);
  CK(mi_scorer_set_receptor(sc, rec_xyz.data(), rec_smt.data(), n_rec));
  std::vector<float> p0(B), a0(B), l0(B), v0(B);
  CK(mi_scorer_score_batch(sc, poses.data(), lig_smt.data(), B, L, nullptr, p0.data(), a0.data(), l0.data(), v0.data()));
  mi_scorer_destroy(sc);
  mi_model_release(m);

  std::printf("devices_visible %d\n", have);
  for (int G = 1; G <= want; G = (G < want && 2 * G > want) ? want : 2 * G) {
    std::vector<int> devs(G);
    for (int g = 0; g < G; g++) devs[g] = g;
    mi_pool *pool = mi_pool_create(devs.data(), G, paths, 1);
    if (!pool) {
      std::fprintf(stderr, "mi_pool_create(%d): %s\n", G, mi_last_error());
      return 4;
    }
    CK(mi_pool_set_receptor(pool, rec_xyz.data(), rec_smt.data(), n_rec));
    std::vector<float> p1(B), a1(B), l1(B), v1(B);
    CK(mi_pool_score_batch(pool, poses.data(), lig_smt.data(), B, L, nullptr, p1.data(), a1.data(), l1.data(), v1.data(), 0));
    const bool host_equal = same_bits(p0, p1) && same_bits(a0, a1) && same_bits(l0, l1);
    const int reps = 5;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++)
      CK(mi_pool_score_batch(pool, poses.data(), lig_smt.data(), B, L, nullptr, p1.data(), a1.data(), l1.data(), v1.data(), 0));
    const double dt_host = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("pool_host devices %d B %d host_path_equal %d poses_per_s %.1f\n", G, B, (int)host_equal, B / dt_host);
    std::fflush(stdout);  // (the host path's result survives whatever the RCCL path below does)
    // (c) everything on devices[0]
    HK(hipSetDevice(0));
    float *d_lig = nullptr, *d_out = nullptr;
    HK(hipMalloc((void **)&d_lig, poses.size() * sizeof(float)));
    HK(hipMalloc((void **)&d_out, (size_t)4 * B * sizeof(float)));
    HK(hipMemcpy(d_lig, poses.data(), poses.size() * sizeof(float), hipMemcpyHostToDevice));
    CK(mi_pool_score_batch(pool, d_lig, lig_smt.data(), B, L, nullptr, d_out, d_out + B, d_out + 2 * B, d_out + 3 * B,
                           MI_LIG_ON_DEVICE | MI_OUT_ON_DEVICE));
    std::vector<float> back((size_t)4 * B);
    HK(hipMemcpy(back.data(), d_out, back.size() * sizeof(float), hipMemcpyDeviceToHost));
    const bool dev_equal = std::memcmp(back.data(), p0.data(), B * sizeof(float)) == 0 &&
                           std::memcmp(back.data() + B, a0.data(), B * sizeof(float)) == 0 &&
                           std::memcmp(back.data() + 2 * B, l0.data(), B * sizeof(float)) == 0;
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++)
      CK(mi_pool_score_batch(pool, d_lig, lig_smt.data(), B, L, nullptr, d_out, d_out + B, d_out + 2 * B, d_out + 3 * B,
                             MI_LIG_ON_DEVICE | MI_OUT_ON_DEVICE));
    const double dt_dev = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("pool devices %d B %d host_path_equal %d poses_per_s %.1f device_path_equal %d poses_per_s %.1f info %s\n", G,
                B, (int)host_equal, B / dt_host, (int)dev_equal, B / dt_dev, mi_pool_info_json(pool));
    std::fflush(stdout);
    HK(hipFree(d_lig));
    HK(hipFree(d_out));
    mi_pool_destroy(pool);
    if (G == want) break;
  }
  return 0;
}
