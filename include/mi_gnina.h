/*
 * mi_gnina.h -- C ABI of the MI355X-native gnina CNN-scoring engine (libmi_gnina.so).
 *
 * gnina has no plugin/FFI loader: its seams are C++ abstract classes compiled into the binary
 * (SURVEY.md §8b).  This header is the boundary a maintainer binds those seams to; every entry
 * point names the reference interface it replaces.  Plain pointers and sizes only, status
 * codes instead of exceptions, caller owns host buffers, library owns device memory.  A handle
 * must not be shared across host threads (the reference's contract too: one scorer copy per
 * thread, gninasrc/main/main.cpp:1438, gninasrc/lib/parallel_mc.cpp:146).
 *
 * Coordinates are float[n][3] Angstrom; atom types are `smt` enum values 0..27
 * (gninasrc/lib/atom_constants.h:45-75) as int32.
 */
#ifndef MI_GNINA_H_
#define MI_GNINA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_GNINA_ABI_VERSION 2 /* 2: mi_ligand_desc gained n_movable / pair_kind / lig_begin / lig_end; mi_cnn_box per_atom_forces */

typedef int mi_status;
enum {
  MI_OK = 0,
  MI_ERR_INVALID = 1,    /* bad argument (reference: internal_error / VINA_CHECK, common.h:280-288) */
  MI_ERR_MODEL = 2,      /* unreadable / unsupported model (reference: usage_error, torch_model.cpp:115-117) */
  MI_ERR_DEVICE = 3,     /* HIP runtime failure */
  MI_ERR_STATE = 4,      /* call order violated (e.g. score before set_receptor) */
  MI_ERR_RANGE = 5       /* mi_scorer_synchronize: a device-output call met an activation outside the fp16 range of the
                            split-fp16 kernels (see MI_PRECISION_FP32); repeat it under MI_PRECISION_FP32_MFMA */
};

typedef struct mi_model mi_model;   /* one network: weights + metadata, device resident, refcounted */
typedef struct mi_scorer mi_scorer; /* one ensemble bound to one receptor; = a DLScorer copy */
typedef struct mi_vina mi_vina;     /* Vina/smina pair-table engine (igrid seam) */

/* flags for the *_ex entry points */
enum {
  MI_MEM_HOST = 0,
  MI_LIG_ON_DEVICE = 1,     /* lig_xyz / centers are device pointers */
  MI_OUT_ON_DEVICE = 2,     /* output arrays are device pointers; call returns after enqueue */
  MI_CENTER_TYPED_ONLY = 4  /* grid centre = mean of TYPED ligand rows only (SURVEY App. A.3 switch) */
};

/* ---- process / thread setup --------------------------------------------------------------
 * Replaces initializeCUDA(int device) (gninasrc/lib/dl_scorer.h:20-21, called main.cpp:753 and
 * parallel_mc.cpp:197): must be called in every host thread that uses the engine.
 * Every scorer / mi_vina handle owns a HIP stream; mi_vina handles driven from different host threads overlap on the
 * device as far as the runtime has hardware queues, so the first call also sets GPU_MAX_HW_QUEUES=16 (HIP's
 * default is 4) unless the variable is already set -- effective only before the process's first HIP call.
 * THREADS.  A scorer (and an mi_vina handle) is driven by one host thread at a time; DIFFERENT scorers of one device may be
 * driven from different host threads at the same time, host-output and MI_OUT_ON_DEVICE calls alike -- that is how gnina
 * drives the seam: fresh_copy() gives every worker thread and every Monte-Carlo task a new CNNTorchScorer with a mutex of
 * its own (cnn_torch_scorer.h:54, dl_scorer.h:43-44, main.cpp:1436-1438, parallel_mc.cpp:145-146).  Every call gives the
 * bits the same call gives alone (tests/test_gpu_concurrency.py: two threads, host and device outputs).  Rounds 5 serialised
 * such calls per device because a voxelizer next to another scorer's conv kernels deviated; round 6 traced that to the
 * voxelizer's packed-fp32 instructions, removed them, and removed the lock (MI_GNINA_CALL_LOCK=1
 * brings it back for A/B measurements).  Within a call of at most 64 poses the models of an ensemble run on streams of their
 * own behind the voxelization (same bits as on one stream; MI_GNINA_LANES=0 switches it off). */
mi_status mi_gnina_init(int device);
/* Experiment / A-B switches (gnina_amd/csrc/options.h lists them: MI_GNINA_*, MI_POOL_*, MI_VINA_*, MI_VOX_*).  The
 * environment is read ONCE per process, at the first library call; afterwards a switch changes only through this call
 * (value NULL = unset) -- nothing in the library calls getenv on a scoring path.  Load-time switches (kernel selection, LDS
 * budgets, fusion) apply to models loaded after the change, run-time ones to the next call.  MI_ERR_INVALID for an unknown
 * name.  mi_gnina_options(): "NAME=value ..." of the switches that are set (thread-local string). */
mi_status mi_gnina_set_option(const char *name, const char *value);
const char *mi_gnina_options(void);
int mi_gnina_device_count(void);
int mi_gnina_abi_version(void);
/* Last error message of the calling thread ("" if none). */
const char *mi_last_error(void);

/* ---- models -------------------------------------------------------------------------------
 * Replaces TorchModel<isCUDA>::TorchModel(std::istream&, name, log) (gninasrc/lib/torch_model.h:32,
 * torch_model.cpp:49-118): `blob` is a MIGNINA1 weight blob -- gnina_amd/tools/extract_weights.py's output for one .pt of
 * gninasrc/lib/models/, or for a user's `--cnn_model file.pt` of another architecture: any TorchScript module whose graph
 * is a feed-forward stack of 2x2x2 / global pools, 1x1x1 / 3x3x3 "same" convolutions, ReLU, eval BatchNorm, channel
 * concatenation and two linear heads is written as the same layer program (the tool names the first operator outside that
 * set; tests/test_extract_generic_cpu.py, tests/test_gpu_custom_model.py).  Returns NULL on failure (see mi_last_error). */
mi_model *mi_model_load(const void *blob, size_t nbytes, const char *name);
mi_model *mi_model_load_file(const char *path);
/* The same weights on another grid: "resolution" / "dimension" as the metadata of a user-supplied model
 * file would give them (torch_model.cpp:73-84); <= 0 keeps the blob's value.  Only networks with a
 * size-independent head qualify (family Dense: dynamic global max pool, e.g. dense_1_3 at 0.25 A / 23.75 A =
 * 96^3); others return NULL (MI_ERR_MODEL). */
mi_model *mi_model_load_file_ex(const char *path, float resolution, float dimension);
void mi_model_retain(mi_model *);
void mi_model_release(mi_model *);
/* TorchModel::get_grid_res / get_grid_dim (torch_model.h:42-43) + channel counts of the two
 * typers (torch_model.h:45-46) + grid points per side (GridMaker::get_first_dim, torch_model.cpp:176). */
mi_status mi_model_info(const mi_model *, float *resolution, float *dimension, int *n_rec_channels,
                        int *n_lig_channels, int *grid_points);
const char *mi_model_name(const mi_model *);
/* FileMappedGninaTyper::get_int_type(smt) as used by make_coordset (torch_model.cpp:120-142):
 * returns the channel index (or -1) and the xs_radius of the original smina type. */
int mi_model_type_channel(const mi_model *, int is_ligand, int smt, float *radius);

/* ---- scorer -------------------------------------------------------------------------------
 * Replaces CNNTorchScorer<isCUDA> (gninasrc/lib/cnn_torch_scorer.h / .cpp:24-92): an ensemble of
 * models.  Creation is cheap (= DLScorer::fresh_copy(), dl_scorer.h:65): models are shared,
 * refcounted and stay device resident. */
mi_scorer *mi_scorer_create(mi_model *const *models, int n_models);
void mi_scorer_destroy(mi_scorer *);
int mi_scorer_num_models(const mi_scorer *);

/* DLScorer::setReceptor (gninasrc/lib/dl_scorer.cpp:93-193): all receptor atoms with their smina
 * types, once; the receptor is assumed constant afterwards (dl_scorer.cpp:112,150).  Types are
 * mapped with each model's receptor typer exactly as make_coordset does for every call in the
 * reference (torch_model.cpp:159). */
mi_status mi_scorer_set_receptor(mi_scorer *, const float *xyz, const int32_t *smt, int n_atoms);

/* Batched form of CNNTorchScorer::score(model&, compute_gradient=false, affinity, loss, variance)
 * (cnn_torch_scorer.cpp:105-198) -> TorchModel::forward (torch_model.cpp:153-224) for B poses of
 * ONE ligand (same atoms/types, different coordinates):
 *   lig_xyz [B][L][3], lig_smt [L]  = DLScorer::setLigand's view of the model (dl_scorer.cpp:36-88)
 *   centers [B][3] or NULL          = cnn_center / NaN -> per-pose ligand mean (torch_model.cpp:163-166)
 *   pose/affinity/loss [B]          = mean over the ensemble of TorchModel::forward's {pose, affinity, loss}
 *   aff_var [B] or NULL             = population variance of affinity over the ensemble
 *                                     (cnn_torch_scorer.cpp:181-191)
 * Synchronous: results are host visible on return (like the reference's score). */
mi_status mi_scorer_score_batch(mi_scorer *, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                                const float *centers, float *pose, float *affinity, float *loss, float *aff_var);
/* Same with MI_* flags: device-resident inputs/outputs, asynchronous on the scorer's stream
 * when MI_OUT_ON_DEVICE is set (use mi_scorer_synchronize). */
mi_status mi_scorer_score_batch_ex(mi_scorer *, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                                   const float *centers, float *pose, float *affinity, float *loss,
                                   float *aff_var, unsigned flags);
/* Arithmetic of the CNN forward pass.  MI_PRECISION_FP32 (default) is the parity path, scores within 1e-4 of the
 * reference: fp32 tensors and fp32 accumulation everywhere; the forward convolutions run on the split-fp16 kernels
 * (each fp32 operand = the sum of two fp16 numbers, three v_mfma_f32_32x32x16_f16 per product, exact products:
 * <= 1e-6 from the fp32-MFMA result on the scores) where a layer has that plan, on fp32 MFMA otherwise.
 * MI_PRECISION_FP32_MFMA forces v_mfma_f32_32x32x2_f32 / 16x16x4_f32 for every layer (the round-2 path).
 * MI_PRECISION_BF16 (BASELINE config 5, "bf16 MFMA path") runs the convolutions on v_mfma_f32_32x32x16_bf16 with bf16
 * activations / weights and fp32 accumulation; voxelization, the fully connected heads and the score post-processing
 * stay fp32.  Its deviation from the fp32 path is a measured tolerance (tests/test_gpu_bf16.py), not the 1e-4 bar.
 * MI_PRECISION_FP16 (round 6: the reduced-precision mode for config 5's Dense network) runs the parity path's own program
 * and tensors, but the Dense family's block layers and transitions (conv3d_h2_d16_kernel / conv3d_h2_k1s_kernel: nine tenths
 * of a Dense pose's FLOPs) issue only the h * h MFMA of every product and neither DMA nor read the l planes: activations and
 * weights rounded to fp16 (11 significant bits against bf16's 8), fp32 accumulation, one MFMA where the parity path issues
 * three.  The other layers keep the parity arithmetic; scoring calls only (gradient calls run the parity path); measured
 * tolerance in tests/test_gpu_bf16.py.
 * Gradient calls (MI_PRECISION_FP32 / _FP32_MFMA; never bf16): the forward half runs the split-fp16 kernels on fp32
 * tensors (the backward pass reads the activations as masks).  For the Default2017 / Default2018 families that is the K
 * order and MFMA order of a scoring call: a pose scores the same bits with and without its gradient.  The Dense family's
 * scoring calls run its block layers on split-format concat buffers with the BatchNorm folded into the weights
 * (conv3d_h2_dense.hip) while gradient calls apply the BatchNorm while staging: a Dense pose's score differs by <= 2e-6
 * between the two kinds of call (tests/test_gpu_gradient.py asserts that bound; 50 times below the parity bar, but not
 * zero: a caller that compares energies of the two kinds of call, like non_cache_cnn::eval vs eval_deriv, sees it).  Under MI_PRECISION_FP32 the
 * 3x3x3 transposed convolutions of the backward pass (and the Dense transitions) run on the split-fp16 kernels as well: a
 * gradient tensor is staged times the power of two that puts its per-pose maximum into [2^14, 2^15) -- recorded by
 * whichever kernel wrote the tensor -- so no magnitude of gradient is out of range, and the atom gradients stay within
 * ~1e-6 of a pose's largest element of what fp32 MFMA returns (tests/test_gpu_gradient.py); the remaining transposed
 * convolutions, and all of them under MI_PRECISION_FP32_MFMA, run on fp32 MFMA.  A pose's gradient does not depend on
 * the other poses of the call.
 * Range of the split-fp16 kernels: the high half of an activation is an fp16 number, so a model (or an input) that
 * produces |activation| > 65504 cannot run on them -- the reference runs any --cnn_model in fp32
 * (torch_model.cpp:49-118,185).  Every split-fp16 kernel and the voxelizer feeding one raise a per-scorer flag when a
 * value they produce or consume leaves that range (NaN included); a call that ends with the flag raised is REPEATED on
 * the fp32-MFMA kernels before it returns (scoring, gradient, CNN-in-the-loop calls alike; mi_scorer_h2_fallbacks
 * counts them).  Device-output calls (MI_OUT_ON_DEVICE) return before their kernels ran: mi_scorer_synchronize then
 * returns MI_ERR_RANGE and the caller repeats them under MI_PRECISION_FP32_MFMA (mi_pool_score_batch does).
 * Small values: an activation below 2^-14 has a subnormal high half, below 2^-24 it is taken as zero -- an absolute
 * error of at most 2^-25 per activation, far below the parity bar for weights of any sane magnitude
 * (tests/test_gpu_h2_range.py drives both ends). */
enum { MI_PRECISION_FP32 = 0, MI_PRECISION_BF16 = 1, MI_PRECISION_FP32_MFMA = 2, MI_PRECISION_FP16 = 3 };
mi_status mi_scorer_set_precision(mi_scorer *, int precision);
/* calls of this scorer that were repeated on the fp32-MFMA kernels because of the range flag (diagnostics, tests) */
int mi_scorer_h2_fallbacks(const mi_scorer *);
/* Test hook (host only): the LDS layout the engine picks for the planar halo tile of the 3x3x3 split-fp16 kernel
 * (conv3d_h2.hip) -- tile of tile_cells3 2x2x2 cells, n_mtiles M-tile slots per workgroup, geometry_mask bit m = M-tile
 * geometry m compiled (0 raster, 1 stacked along x, 2 a 2 x 2 square in (x, y)) -> {geometry, pad slots per z-row, per
 * x-plane} and the bank model's mean LDS cycles per ds_read_b128 lane group (1.0 = conflict free). */
mi_status mi_debug_h2_layout(const int32_t *tile_cells3, int n_mtiles, int geometry_mask, int32_t *mt_pad_y_pad_x, float *cycles);
/* Diagnostic, host only (no device needed): the operand split of the split-fp16 kernels as the model loader applies it
 * to the weights -- hi[i] = RN_fp16(x[i] * scale), lo[i] = RN_fp16(x[i] * scale - hi[i]) as IEEE binary16 bit patterns
 * (round to nearest even, subnormals kept).  scale = 0 picks the per-layer power of two the loader would: the one that
 * lifts max |x| into [2^13, 2^14); the scale used is returned in *scale_out (may be NULL). */
mi_status mi_debug_split_f16(const float *x, int n, float scale, uint16_t *hi, uint16_t *lo, float *scale_out);
/* Diagnostic: activation buffer `buf` (the model blob's "buf" ids) of model `model_index` as the LAST forward call of this
 * scorer left it, decoded to fp32 channels-last [B][S][S][S][C] on the host -- a split-format buffer (h + l) as well as an
 * fp32 one.  info[3] = {S, C, 1 if the buffer is in the split format under the scorer's precision}; out may be NULL to
 * query info only.  tests/test_gpu_dense_split.py compares the split-fp16 and fp32-MFMA programs layer by layer with it. */
mi_status mi_debug_read_activation(mi_scorer *, int model_index, int buf, int B, int32_t *info, float *out, size_t out_floats);
/* Diagnostic: the candidate lists gather_pose_atoms left for pose 0 of the last call (what voxelize_tiles read): info[2] =
 * {n_slab, cap}; counts[n_slab]; chan[n_slab][cap] (channel of every candidate, ascending per list); rec[n_slab][cap][8]
 * (x y z radius and the density constants).  Pass NULL arrays to query info only.  tools/experiments/concurrency_diag3.py. */
mi_status mi_debug_read_candidates(mi_scorer *, int32_t *info, int32_t *counts, int32_t *chan, float *rec);
/* Diagnostic (tools/experiments/vox_stress.py; round 6's hunt for what a second hardware queue does to the voxelizer):
 * gather + voxelize ONE pose `iters` times on the scorer's stream, nothing else, and compare every iteration's pooled grid on
 * the device with the grid a call with flag 1 recorded.  flags: 1 = record the reference grid and return; 2 = gather only
 * before the first iteration; 4 = fill the grid with 0xFF bytes before every iteration; 8 = hand the tile kernel a trap
 * ring (a -DMI_VOX_TRAP build of voxelize.hip reports into it) and copy it to trap_out [1024][16].  log [log_cap][4]: row 0 =
 * {differing dwords, 0, compare workgroups that saw one, 0}, then {iteration, dword index, got, want}. */
mi_status mi_debug_vox_stress(mi_scorer *, const float *lig_xyz, const int32_t *lig_smt, int L, int iters, int flags,
                              int32_t *log, int log_cap, uint32_t *trap_out);
/* Virtual screening (1 receptor x many ligands): B poses that may each belong to a
 * different ligand, in one batch.  lig_xyz [B][Lmax][3], lig_smt [B][Lmax]: pose b's atoms are the leading
 * rows with smt >= 0, the remaining rows are padding (smt = -1, coordinates ignored).  Everything else as
 * mi_scorer_score_batch (one DLScorer::setLigand + score per pose in the reference, main.cpp:1438-1466).
 * Host pointers. */
mi_status mi_scorer_score_ragged(mi_scorer *, const float *lig_xyz, const int32_t *lig_smt, int B, int Lmax,
                                 const float *centers, float *pose, float *affinity, float *loss, float *aff_var);
/* CNNTorchScorer::score(model&, compute_gradient = true, ...) for B poses: forward, loss.backward()
 * through the network and GridMaker::backward (torch_model.cpp:197-221; cnn_torch_scorer.cpp:164-175).
 * lig_grad [B][L][3] = d loss / d x for every ligand row (0 for untyped rows such as hydrogens), mean
 * over the ensemble -- what getGradient + add_minus_forces + scale_minus_forces(1/cnt) leave in
 * model::minus_forces.  Host pointers.  Supported for the Default2017 / Default2018 / Dense families and the
 * Overlap test model (mi_model_supports_gradient). */
mi_status mi_scorer_score_grad(mi_scorer *, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                               const float *centers, float *pose, float *affinity, float *loss, float *aff_var,
                               float *lig_grad);
int mi_model_supports_gradient(const mi_model *);

/* Flexible receptor residues.  DLScorer::setReceptor keeps the movable receptor atoms in the first rows of
 * receptor_coords and refreshes them from the model on every call (dl_scorer.cpp:150-193); their gradient
 * travels back through receptor_map (getReceptorGradient, cnn_torch_scorer.cpp:216-224).
 * mi_scorer_set_flex names the rows of mi_scorer_set_receptor's arrays that move (any rows, n_flex >= 0;
 * a later mi_scorer_set_receptor clears the declaration).  mi_scorer_score_flex scores B poses with
 * per-pose coordinates flex_xyz [B][n_flex][3] for those rows (NULL: the stored coordinates); lig_grad
 * [B][L][3] and flex_grad [B][n_flex][3] are optional outputs (both NULL: forward only).  Gradients of
 * untyped rows are 0.  Host pointers. */
mi_status mi_scorer_set_flex(mi_scorer *, const int32_t *rec_rows, int n_flex);
int mi_scorer_flex_count(const mi_scorer *); /* rows declared by the last mi_scorer_set_flex */
/* TorchModel::forward's `rotate` (torch_model.cpp:170-173,204-206; used for --cnn_rotation averaging,
 * cnn_torch_scorer.cpp:130-141): libmolgrid's Transform(gcenter, 0, rotate) turns every atom -- receptor and ligand --
 * about the grid centre, x' = R(q)(x - c) + c, before GridMaker::forward, and Transform::backward turns the atom
 * gradients back.  quats [B][4] = unit quaternions (a, b, c, d), one per pose ((1,0,0,0) = none); they apply to the
 * NEXT scoring call of this scorer only (score_batch / score_ragged / score_grad / score_flex / voxelize_batch),
 * whose B must match.  Which quaternions libmolgrid draws is the caller's business (HipCNNScorer, hip_cnn_scorer.cpp). */
mi_status mi_scorer_set_rotations(mi_scorer *, const float *quats, int B);
mi_status mi_scorer_score_flex(mi_scorer *, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                               const float *centers, const float *flex_xyz, float *pose, float *affinity,
                               float *loss, float *aff_var, float *lig_grad, float *flex_grad);

/* Per-model raw outputs of the last mi_scorer_score_batch* call: TorchModel::forward's
 * {pose, affinity, loss} for model `m` (host arrays [B]); used by the parity tests. */
mi_status mi_scorer_last_model_outputs(mi_scorer *, int m, float *pose, float *affinity, float *loss, int B);

/* GridMaker::forward as invoked at torch_model.cpp:175-181 (and by gninagrid, molgridder.cpp:100-138)
 * for model `m` of the ensemble: writes float [B][C][N][N][N] (x slowest, z fastest), zero filled
 * first.  grid_out is host memory unless flags has MI_OUT_ON_DEVICE.  centers_out [B][3] (host,
 * optional) receives the grid centres used. */
mi_status mi_voxelize_batch(mi_scorer *, int m, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                            const float *centers, float *grid_out, float *centers_out, unsigned flags);

/* CNN forward only, on caller-provided grids in the reference layout [B][C][N][N][N]
 * (= module.forward, torch_model.cpp:185, plus the post-processing of :188-195). Host pointers. */
mi_status mi_model_forward_grids(mi_scorer *, int m, const float *grids, int B, float *pose, float *affinity,
                                 float *loss);

/* Stream plumbing for benchmarks: the hipStream_t all work of this scorer is enqueued on. */
void *mi_scorer_stream(mi_scorer *);
mi_status mi_scorer_synchronize(mi_scorer *);

/* ---- one node, many GPUs, one process ---------------------------------------------------------------
 * Replaces the worker / thread-pool fan-out of gnina (one DLScorer::fresh_copy() per worker thread, main.cpp:1418-1442;
 * parallel_mc.cpp:183-214) by a fan-out over devices: mi_pool owns one host thread, one stream and one mi_scorer per
 * listed GPU -- model weights and the receptor are replicated (a few MB) -- and splits a batch of B poses into
 * contiguous shards [g B/G, (g+1) B/G) that run concurrently.  Poses are independent: no collective on the data path;
 * results land in the caller's arrays in pose order and equal, bit for bit, what one mi_scorer returns for the batch.
 *   host buffers    every worker moves its own shard over its own PCIe link (no GPU-to-GPU traffic);
 *   MI_LIG_ON_DEVICE | MI_OUT_ON_DEVICE
 *                   poses / centres / outputs live on devices[0]: shards are scattered and the scores gathered over
 *                   xGMI with RCCL (ncclSend / ncclRecv groups); librccl.so is opened on first use of this path only
 *                   (MI_POOL_NO_RCCL, or a librccl that cannot be initialised: plain device-to-device copies instead).
 * A pool of one device forwards to its single scorer (any flag combination).  Not thread-safe: one caller at a time. */
typedef struct mi_pool mi_pool;
mi_pool *mi_pool_create(const int *devices, int n_devices, const char *const *model_paths, int n_models);
void mi_pool_destroy(mi_pool *);
int mi_pool_size(const mi_pool *);
mi_status mi_pool_set_receptor(mi_pool *, const float *xyz, const int32_t *smt, int n_atoms);
/* mi_scorer_score_batch_ex over the pool's devices (flags as above) */
mi_status mi_pool_score_batch(mi_pool *, const float *lig_xyz, const int32_t *lig_smt, int B, int L, const float *centers,
                              float *pose, float *affinity, float *loss, float *aff_var, unsigned flags);
/* mi_scorer_score_ragged over the pool's devices: the virtual-screen seam (config C4), host buffers */
mi_status mi_pool_score_ragged(mi_pool *, const float *lig_xyz, const int32_t *lig_smt, int B, int Lmax,
                               const float *centers, float *pose, float *affinity, float *loss, float *aff_var);
/* {"devices": [...], "ranks": G, "rccl_loaded": bool, "rccl_comms": bool, "calls_host_path": n, "calls_device_path": n} */
const char *mi_pool_info_json(mi_pool *);
/* Max poses processed per internal chunk (activation workspace is sized for it). */
mi_status mi_scorer_set_chunk(mi_scorer *, int poses_per_chunk);
/* Per-stage device time of the last score call, milliseconds, measured with HIP events on the
 * scorer's stream: [0] gather+voxelize, [1] CNN convs+heads, [2] total.  Requires
 * mi_scorer_enable_timing(s, 1) before the call. */
mi_status mi_scorer_enable_timing(mi_scorer *, int on);
mi_status mi_scorer_last_timing(mi_scorer *, float *ms3);

/* ---- Vina / smina scoring + local optimisation (igrid / quasi_newton seam) -------------------------
 * mi_vina owns: the pair tables of precalculate_linear(sf, factor) (gninasrc/lib/precalculate.h:165-272;
 * selection main.cpp:1384-1391) for the default scoring function (main.cpp:1324-1329) or custom
 * weights {gauss1, gauss2, repulsion, hydrophobic, h-bond}; the receptor atoms; the per-ligand-type
 * affinity grids of `cache` (cache.cpp:104-184); and one prepared ligand (torsion tree + pairs). */
mi_vina *mi_vina_create(const float *weights5 /* NULL = default Vina */, float cutoff /* 8 */, float factor /* 32 */);
void mi_vina_destroy(mi_vina *);
/* table points per type pair: sz(factor * cutoff^2) + 3 (precalculate.h:189) */
int mi_vina_table_size(const mi_vina *);
/* precalculate_linear_element data of one type pair: fast[n], smooth.first[n], smooth.second[n] */
mi_status mi_vina_table(const mi_vina *, int t1, int t2, float *fast, float *smooth_e, float *smooth_dor);
/* model.grid_atoms: the rigid receptor, all atoms with smina types (hydrogens included) */
mi_status mi_vina_set_receptor(mi_vina *, const float *xyz, const int32_t *smt, int n_atoms);
/* cache::populate(m, prec, atom_types_needed, ...) on the grid_dims {begin, end, n} of
 * setup_grid_dims (main.cpp:625-634); slope = out-of-box penalty slope (main.cpp:466). */
mi_status mi_vina_build_cache(mi_vina *, const float *begin3, const float *end3, const int32_t *n3,
                              const int32_t *lig_types, int n_types, float slope);
/* copy one type's grid back: float[(n3[2]+1)][(n3[1]+1)][(n3[0]+1)], x fastest (array3d.h:91-96) */
mi_status mi_vina_cache_grid(mi_vina *, int smt, float *out, size_t n_floats);

/* --user_grid (main.cpp:993,1342-1350): a caller-supplied potential on its own lattice.
 * mi_user_grid_parse = setup_user_gd (main.cpp:635-670) + the value lines of grid::init(gd, user_in, scale)
 * (grid.cpp:69-92): three header lines, "SPACING g", "NELEMENTS nx ny nz", "CENTER cx cy cz", then one number per
 * line, x fastest.  Call with values = NULL to learn *n_values (= nx' ny' nz' of the derived grid_dims), then again
 * with room for them.  mi_vina_set_user_grid stores -(value * scaling_factor) like the reference (values = NULL
 * removes the grid) and must precede mi_vina_build_cache to take part in the cache grids.  What it changes, exactly
 * as in the reference: cache::populate adds evaluate_user at every lattice point -- with the point's INDICES as the
 * location (cache.cpp:177-179); non_cache::eval_deriv (MI_VINA_DIRECT with derivatives: refine_structure) and
 * non_cache_cnn::eval_deriv (mi_cnn_eval_batch / mi_cnn_refine_batch) add it per heavy atom at the atom's
 * coordinates (non_cache.cpp:168-173, non_cache_cnn.cpp:141-150); the igrids' energy-only evaluations do not see it
 * (non_cache.cpp:76 has the term commented out), but model::eval (mi_vina_eval_batch with_deriv = 0, hence
 * mi_vina_final_energies) adds its own sum over ALL atoms of the ligand at slope 1000 (model.cu:125-134). */
/* --approximation linear | spline (main.cpp:904-905,989-990,1384-1391): the `precalculate` behind every pair term of
 * this handle -- cache::populate, model::eval / eval_deriv, non_cache, the minimiser and the search.  LINEAR =
 * precalculate_linear(sf, 32), what mi_vina_create builds; SPLINE = precalculate_splines(sf, factor)
 * (precalculate.h:277-449, splines.h): a clamped cubic spline of E(r) per type pair over factor * cutoff intervals --
 * gnina's default for --minimize, with factor 10 (main.cpp:1162-1165).  Call before mi_vina_build_cache (a cache
 * built under the other approximation is dropped).  MI_VINA_EXACT in a with_deriv argument still selects
 * precalculate_exact for that call. */
enum { MI_VINA_APPROX_LINEAR = 0, MI_VINA_APPROX_SPLINE = 1 };
/* --accurate_line_search (minimization_params::BFGSAccurateLineSearch, bfgs.h:395-400): kind 1 makes every BFGS of
 * this handle -- mi_vina_bfgs_batch, mi_vina_refine_*, the Monte-Carlo chains, mi_cnn_refine_batch and the CNN
 * Monte-Carlo -- use accurate_line_search (bfgs.h:104-180, backtracking with cubic interpolation after Numerical
 * Recipes' lnsrch) instead of fast_line_search; 0 (default) returns to the fast one.  kind 2 = --simple_ascent
 * (minimization_params::Simple, quasi_newton.cpp:77-79): simple_gradient_ascent (bfgs.h:234-355), steepest descent under
 * the accurate line search with no quasi-Newton update.  (The enum's fourth value, ConjugateGradient, has no
 * implementation in the reference either: quasi_newton.cpp runs bfgs<> for it.) */
mi_status mi_vina_set_line_search(mi_vina *, int kind);
/* Strict summation order (on = 1): every energy sum of this handle's evaluations -- cache::eval / eval_deriv over the
 * atoms (cache.cpp:52-83), non_cache over the receptor atoms (non_cache.cpp:52-83,125-179), the interacting pairs
 * (model.cu:22-60) -- is added in the reference's order instead of a wavefront butterfly.  Forces, coordinates and
 * sinf / cosf already follow the reference bit for bit in either mode; with strict order the energies do too, so BFGS
 * runs and Monte-Carlo chains reproduce the reference's trajectories exactly (about 1 us more per evaluation; default
 * 0).  eval_intramolecular of a model with flexible residues (with_deriv = 4, the `intramolecular` of
 * mi_vina_final_energies) adds its flex-rigid and flex-flex terms one by one like model.cu:352-399. */
mi_status mi_vina_set_strict_order(mi_vina *, int on);
/* Diagnostic: sn[i], cs[i] = the device's sinf(x[i]), cosf(x[i]) as the Vina kernels compute them -- glibc's
 * algorithm restated in fp64 so that tree.h / quaternion.h's std::sin / std::cos give the host's bits (|x| < 120). */
mi_status mi_debug_sincos(const float *x, int n, float *sn, float *cs);
/* ex[i] = the device's expf(x[i]) (Metropolis criterion, precalculate_exact), lg[i] = its logf(|x[i]|) (random_normal),
 * likewise glibc's algorithms restated; logf for positive normal arguments. */
mi_status mi_debug_explog(const float *x, int n, float *ex, float *lg);
/* ac[i] = the device's acosf(x[i]) (quaternion_to_angle inside the accurate line search): fdlibm's fp32 algorithm. */
mi_status mi_debug_acos(const float *x, int n, float *ac);
mi_status mi_vina_set_approximation(mi_vina *, int kind, float factor);
/* precalculate::eval_deriv(a, b, r2) of the current approximation for one type pair: e[i], dor[i] = (E, (dE/dr) / r) at
 * r2[i] (host arrays; linear: r2 <= cutoff^2). */
mi_status mi_vina_pair_eval(mi_vina *, int t1, int t2, const float *r2, int n, float *e, float *dor);
mi_status mi_user_grid_parse(const char *text, size_t len, float begin[3], float end[3], int32_t n[3], double *values,
                             size_t cap, size_t *n_values);
mi_status mi_vina_set_user_grid(mi_vina *, const float begin[3], const float end[3], const int32_t n[3],
                                const double *values, float scaling_factor);

/* model.ligands[0] as the PDBQT parser lays it out (parse_pdbqt.cpp:343-380; tree.h:152-233):
 * nodes in DFS pre-order, node 0 = rigid root, node k>0 = segment owning torsion k-1; atoms stored
 * node by node with coordinates local to their node; interacting pairs per model::initialize_pairs
 * (model.cpp:682-703).  conf = [position 3][orientation quaternion a,b,c,d][torsions n_nodes-1];
 * change = [force 3][torque 3][torsion derivatives] (conf.h:244-359,361-518). */
typedef struct mi_ligand_desc {
  int32_t n_atoms;
  const int32_t *smt;             /* [n_atoms] */
  const float *local_xyz;         /* [n_atoms][3] */
  int32_t n_nodes;
  const int32_t *node_parent;     /* [n_nodes], -1 for the root */
  const int32_t *node_atom_begin; /* [n_nodes] */
  const int32_t *node_atom_end;   /* [n_nodes] */
  const float *node_rel_origin;   /* [n_nodes][3] segment::relative_origin (root: unused) */
  const float *node_rel_axis;     /* [n_nodes][3] segment::relative_axis */
  int32_t n_pairs;
  const int32_t *pairs;           /* [n_pairs][2] */
  /* Flexible receptor residues in the search (model.h: atoms = [flex movable | ligand | inflex]; flex.derivative
   * tree.h:374-393; model::other_pairs model.cu:209-213; conf = [7 + T_ligand + T_flex], conf.h:361-373).  All zero /
   * NULL for a plain ligand.  mi_pdbqt_model_open builds such a description from .pdbqt files.
   *   node_parent[k] == -2: node k is a residue's first_segment -- it hangs off the world, node_rel_origin / axis are
   *                         absolute; its subtree follows it like a ligand branch;
   *   atoms outside every node: inflex atoms, fixed at local_xyz (absolute), partners in pairs only;
   *   n_movable:  atoms [0, n_movable) get the receptor term (0 = n_atoms);
   *   pair_kind:  [n_pairs] 0 = ligand-internal pair (curl cap v[0]), 1 = pair of model::other_pairs (cap v[2]);
   *   lig_begin / lig_end: the ligand's atom range (mutate_conf's gyration radius); 0, 0 = all atoms. */
  int32_t n_movable;
  const int32_t *pair_kind;
  int32_t lig_begin, lig_end;
} mi_ligand_desc;
mi_status mi_vina_set_ligand(mi_vina *, const mi_ligand_desc *);
/* model::eval_deriv(p, ig = cache, v, conf, change) (model.cu:202-225) [with_deriv = 1] or model::eval
 * (energy only, the Metropolis energy of monte_carlo.cpp:44-47) for B conformations.
 * v3 = per-term positive-energy caps {intramolecular, receptor grid, other} (curl.h:29-42; hunt cap
 * (10,10,10) / authentic (1000,1000,1000), main.cpp:460, monte_carlo.cpp:102).
 * confs [B][7+T]; energy [B]; change [B][6+T] or NULL; coords [B][n_atoms][3] or NULL. */
mi_status mi_vina_eval_batch(mi_vina *, const float *confs, int B, const float *v3, int with_deriv, float *energy,
                             float *change, float *coords);
/* with_deriv values: 1 model::eval_deriv, 0 model::eval, 2 receptor term only (cache::eval / non_cache::eval),
 * 4 eval_intramolecular (ligand pairs only, model.cu:352-399); OR-able flags: */
enum {
  MI_VINA_DIRECT = 16, /* receptor term from the atoms, no grids: the non_cache igrid (non_cache.cpp:52-83,125-179) */
  MI_VINA_EXACT = 32,  /* precalculate_exact instead of the linear tables (precalculate.h:452-494) */
  MI_VINA_USER_TERM = 64 /* with_deriv = 2 only: add model::eval's user-grid sum over the ligand's atoms (model.cu:125-134) */
};
/* refine_structure (main.cpp:131-171): BFGS on non_cache, out-of-box slope 10 -> x10 per try (<= 5) until
 * non_cache::within; energy = max_fl when the pose never gets inside.  In place; tries [B] optional. */
mi_status mi_vina_refine_batch(mi_vina *, float *confs, int B, const float *v3, int max_iters, float *energy,
                               int32_t *tries);
/* The energies do_search's docking branch reports (main.cpp:339-344): intramolecular = eval_intramolecular(exact_prec);
 * e_final = conf_independent(eval(exact_prec, nc_new) - intramolecular) with the default num_tors_div weight
 * (everything.h:796-814, main.cpp:1329).  nc_new is a non_cache on the run's LINEAR tables (main.cpp:231), so the
 * receptor term is the table look-up (precalculate::eval = eval_fast) and only the pair terms are exact.
 * num_tors = conf_independent_inputs::num_tors (terms.cpp:74-106; mi_pdbqt_ligand_num_tors), not TORSDOF. */
mi_status mi_vina_final_energies(mi_vina *, const float *confs, int B, const float *v3, float num_tors,
                                 float *e_final, float *intramolecular);
/* quasi_newton::operator() (quasi_newton.cpp:49-83) = bfgs<> with fast_line_search (bfgs.h:73-91,
 * 357-502) for B conformations, in place; max_iters = (25 + n_movable_atoms) / 3 in gnina
 * (main.cpp:454-456).  energy [B]; grad [B][6+T] or NULL; evals [B] or NULL. */
mi_status mi_vina_bfgs_batch(mi_vina *, float *confs, int B, const float *v3, int max_iters, float *energy,
                             float *grad, int32_t *evals);
/* monte_carlo::operator() (monte_carlo.cpp:99-148) for B independent chains of the current ligand:
 * random start in the box (conf.h:119-122), per step mutate_conf (mutate.cpp:35-73) -> BFGS with the
 * hunt cap -> Metropolis on the receptor-grid energy (monte_carlo.cpp:38-47) -> BFGS with the full cap
 * -> add_to_output_container (coords.cpp:43-56).  Defaults of gnina: temperature 1.2, amplitude 2,
 * min_rmsd 1.0, num_saved 50, hunt_cap (10,10,10), authentic_v (1000,1000,1000), n_steps per
 * main.cpp:441-463.  One seed per chain (parallel_mc.cpp:190-192).  The random stream is NOT
 * boost::mt19937 (unvendored dependency): parity with the reference is statistical.
 * Outputs per chain, sorted by energy: out_n [B]; out_e [B][num_saved]; out_conf [B][num_saved][7+T];
 * out_coords [B][num_saved][n_heavy][3] (heavy-atom coordinates, model::get_heavy_atom_movable_coords). */
typedef struct mi_mc_params {
  int32_t n_steps, max_iters, num_saved;
  float temperature, mutation_amplitude, min_rmsd;
  float hunt_cap[3], authentic_v[3];
} mi_mc_params;
mi_status mi_vina_mc_batch(mi_vina *, int B, const uint64_t *seeds, const float *corner1, const float *corner2,
                           const mi_mc_params *params, int32_t *out_n, float *out_e, float *out_conf,
                           float *out_coords, int32_t *evals);
int mi_vina_ligand_heavy_atoms(const mi_vina *);
/* Virtual screening: MANY ligands docked by ONE launch.  gnina docks the ligands of a screen one after the other,
 * `exhaustiveness` chains each on a thread pool (main.cpp:1001-1075 ligand loop, parallel_mc.cpp:183-214); one
 * ligand's 8 chains occupy a sliver of the device and streams of different handles only overlap as far as the
 * runtime has hardware queues, so the screen's chains are put into one grid instead: chain b docks ligand
 * chain_ligand[b] of the set given to mi_vina_set_screen, with that ligand's own n_steps / max_iters
 * (params[l], l < n_lig; num_saved must agree; temperature, amplitude, min_rmsd and the caps are taken from
 * params[0]).  Same search box, receptor and cache grids for all (build the cache for the union of the ligands'
 * atom types).  Outputs as mi_vina_mc_batch with common strides: out_conf [B][num_saved][max_conf], out_coords
 * [B][num_saved][max_heavy][3] (mi_vina_screen_dims); a chain of ligand l fills the first 7+T_l / n_heavy_l*3
 * floats of its rows.  A chain's result is bit-identical to mi_vina_mc_batch of that ligand with the same seed.
 * Chains may be listed in any order: the launch hands them to the device longest search first. */
mi_status mi_vina_set_screen(mi_vina *, int n_lig, const mi_ligand_desc *descs);
int mi_vina_screen_size(const mi_vina *);
mi_status mi_vina_screen_dims(const mi_vina *, int32_t *max_conf, int32_t *max_heavy, int32_t *max_atoms);
mi_status mi_vina_mc_screen(mi_vina *, int B, const int32_t *chain_ligand, const uint64_t *seeds, const float *corner1,
                            const float *corner2, const mi_mc_params *params, int32_t *out_n, float *out_e,
                            float *out_conf, float *out_coords, int32_t *evals);
/* The per-ligand tail of do_search (main.cpp:324-347) for the poses of a whole screen in one launch each: item b
 * is a conformation of ligand item_ligand[b].  Rows are strided by the maxima over the set (mi_vina_screen_dims):
 * confs [B][max_conf], change [B][max_conf-1], coords [B][max_atoms][3]; an item of ligand l uses the first
 * 7+T_l / 6+T_l / 3*n_atoms_l floats of its rows.  Same semantics and bits as mi_vina_eval_batch /
 * mi_vina_refine_batch / mi_vina_final_energies of that ligand; max_iters and num_tors are per ligand [n_lig]. */
mi_status mi_vina_eval_screen(mi_vina *, const int32_t *item_ligand, const float *confs, int B, const float *v3,
                              int with_deriv, float *energy, float *change, float *coords);
mi_status mi_vina_refine_screen(mi_vina *, const int32_t *item_ligand, float *confs, int B, const float *v3,
                                const int32_t *max_iters, float *energy, int32_t *tries);
mi_status mi_vina_final_energies_screen(mi_vina *, const int32_t *item_ligand, const float *confs, int B,
                                        const float *v3, const float *num_tors, float *e_final, float *intramolecular);
/* ---- the same for the Vina / Monte-Carlo half of the path -------------------------------------------------------------
 * Replaces parallel_mc's fan-out of chains over a thread pool (parallel_mc.cpp:183-214: one task per chain, each with
 * its own model copy and seed; containers merged afterwards, :165-181) by a fan-out over devices.  mi_vina_pool owns one
 * host thread and one mi_vina handle per listed GPU.  Set-up is REPLICATED: mi_vina_pool_configure runs fn(handle, rank,
 * user) once per device, on that device's thread -- call mi_vina_set_receptor / mi_vina_build_cache / mi_vina_set_ligand
 * (or mi_vina_set_screen) / option setters in it, identically for every rank; the cache grids are built per device
 * (a few MB each; faster than shipping them).  mi_vina_pool_mc_batch = mi_vina_mc_batch with the B chains split by chain
 * id into contiguous shards; mi_vina_pool_mc_screen = mi_vina_mc_screen with ligand l's chains on device l % G.  conf_size
 * = 7 + T (+ flexible-residue torsions), n_heavy = mi_vina_ligand_heavy_atoms; max_conf / max_heavy = mi_vina_screen_dims.
 * A chain depends on its seed and the handle's state only: the outputs equal those of one handle bit for bit for any
 * number of devices.  Host arrays in, host arrays out; merge with mi_merge_mc_outputs.  No collective. */
typedef struct mi_vina_pool mi_vina_pool;
mi_vina_pool *mi_vina_pool_create(const int *devices, int n_devices, const float *weights5, float cutoff, float factor);
void mi_vina_pool_destroy(mi_vina_pool *);
int mi_vina_pool_size(const mi_vina_pool *);
typedef int (*mi_vina_pool_fn)(mi_vina *handle, int rank, void *user); /* returns an mi_status (MI_OK = 0) */
mi_status mi_vina_pool_configure(mi_vina_pool *, mi_vina_pool_fn fn, void *user);
mi_status mi_vina_pool_mc_batch(mi_vina_pool *, int B, const uint64_t *seeds, const float *corner1, const float *corner2,
                                const mi_mc_params *params, int conf_size, int n_heavy, int32_t *out_n, float *out_e,
                                float *out_conf, float *out_coords, int32_t *evals);
mi_status mi_vina_pool_mc_screen(mi_vina_pool *, int B, const int32_t *chain_ligand, const uint64_t *seeds,
                                 const float *corner1, const float *corner2, const mi_mc_params *params, int max_conf,
                                 int max_heavy, int32_t *out_n, float *out_e, float *out_conf, float *out_coords,
                                 int32_t *evals);
const char *mi_vina_pool_info_json(mi_vina_pool *);
/* do_search's ranking tail (main.cpp:348-361): sort (pose_sort_order: CNNscore / CNNaffinity descending,
 * Energy ascending) then remove_redundant(out_cont, out_min_rmsd) (main.cpp:182-192).  Host only.
 * coords [n_poses][n_heavy][3]; order_out [n_poses] receives the kept pose indices, best first. */
/* merge_output_containers (parallel_mc.cpp:165-181): fold the per-chain containers of mi_vina_mc_batch
 * into one (add_to_output_container with min_rmsd, gnina uses 2.0, and max_size), sorted by energy.  Host only. */
mi_status mi_merge_mc_outputs(const int32_t *in_n, const float *in_e, const float *in_conf, const float *in_coords,
                              int B, int S, int conf_len, int n_heavy, float min_rmsd, int max_size, int32_t *out_n,
                              float *out_e, float *out_conf, float *out_coords);
enum { MI_SORT_CNNSCORE = 0, MI_SORT_CNNAFFINITY = 1, MI_SORT_ENERGY = 2 };
mi_status mi_rank_poses(const float *cnnscore, const float *cnnaffinity, const float *energy, const float *coords,
                        int n_poses, int n_heavy, int sort_order, float min_rmsd, int32_t *order_out,
                        int32_t *n_out);
/* ---- CNN in the optimisation loop (--cnn_scoring refinement and above) ---------------------------------
 * non_cache_cnn (non_cache_cnn.cpp:33-54,79-169) is the igrid quasi_newton minimises on (main.cpp:475-476):
 *   E(conf) = CNN loss of the ensemble + slope * (distance of every heavy movable atom outside the search
 *             box gd + outside the CNN cube cnn_gd);   no intramolecular term (skip_interacting_pairs()),
 *   minus_forces = ensemble-mean d loss / d x of the heavy atoms + the penalty forces; hydrogens 0.
 * The ligand of mi_vina_set_ligand supplies the torsion tree; ALL of its atoms (hydrogens included, as
 * DLScorer::setLigand does) go to the CNN with their smina types; the receptor is the scorer's.
 * Every evaluation re-centres the CNN grid on the ligand (cnn_torch_scorer.cpp:137-142); the penalty cube
 * cnn_gd is given by cnn_centers [B][3] (NULL: inactive, as before adjust_center) and box->cnn_dimension. */
typedef struct mi_cnn_box {
  int32_t use_search_box;        /* gd[j].n > 0 */
  float box_begin[3], box_end[3];
  float cnn_dimension;           /* DLScorer::set_bounding_box: the model's grid dimension (23.5) */
  float slope;                   /* mi_cnn_eval_batch only; mi_cnn_refine_batch runs refine_structure's ladder */
  /* cnn_options::mix_emp_force / mix_emp_energy / empirical_weight (user_opts.h:46-51; non_cache_cnn.cpp:113-166):
   * blend the empirical receptor term (precalculate_linear pair tables vs the mi_vina receptor, at the atom's
   * coordinates clamped to the search box, curl cap v) into the force and / or the energy.  As in the reference
   * the empirical term is only evaluated when mix_emp_force is set, and ::eval (with_deriv = 0) ignores both. */
  int32_t mix_emp_force, mix_emp_energy;
  float empirical_weight;        /* default 1 */
  float v;                       /* authentic_v[1] = 1000 */
  /* How the CNN's atom gradient reaches model::minus_forces.  0 (default) = as the reference does it:
   * CNNTorchScorer::getGradient fills one entry per movable atom, hydrogens included (cnn_torch_scorer.cpp:208-228), and
   * model::add_minus_forces (model.cu:247-259) walks the movable atoms with a counter that advances only on
   * non-hydrogen atoms -- the k-th heavy atom receives entry k.  For a ligand whose hydrogens all follow its heavy
   * atoms that is every atom's own gradient; with polar hydrogens in between (the usual PDBQT order) heavy atoms
   * behind the first hydrogen receive a neighbour's.  gnina's own code running on HipCNNScorer through the DLScorer seam
   * (tests/cpp/test_cnn_dropin.cpp) does exactly this, so the batched entry points default to it.
   * 1 = every heavy atom its own gradient (what the formula intends; NOT what gnina computes). */
  int32_t per_atom_forces;
} mi_cnn_box;
/* The `igrid` seam itself (igrid.h:32-46) for the cache igrid: cache::eval (minus_forces = NULL) / cache::eval_deriv
 * (cache.cpp:50-83) on B coordinate sets coords [B][n_atoms][3] of atoms with smina types smt [n_atoms] -- what a caller
 * holding a gnina `model` passes (m.coords, movable atoms).  energy [B]; minus_forces [B][n_atoms][3] receives
 * what cache::eval_deriv leaves in model::minus_forces (0 for hydrogens).  v = the curl cap (v[1]).  Host pointers. */
mi_status mi_vina_cache_eval_coords(mi_vina *, const float *coords, const int32_t *smt, int n_atoms, int B, float v,
                                    float *energy, float *minus_forces);
/* model::set(conf) for B conformations: coords [B][n_atoms][3]. */
mi_status mi_vina_coords_batch(mi_vina *, const float *confs, int B, float *coords);
/* non_cache_cnn::eval_deriv (with_deriv = 1: energy [B], change [B][6+T]) / ::eval (0: energy only). */
mi_status mi_cnn_eval_batch(mi_vina *, mi_scorer *, const float *confs, int B, const mi_cnn_box *box,
                            const float *cnn_centers, int with_deriv, float *energy, float *change);
/* refine_structure (main.cpp:131-171) with nc = non_cache_cnn for B poses at once: adjust_center (cube
 * centred on the heavy atoms of the starting pose), then quasi_newton with the slope ladder 10, 100, ...
 * (<= 5 tries) until non_cache_cnn::within; energy = max_fl if the pose never gets inside.  The optimiser
 * state (bfgs.h:357-502) advances on the host; the evaluations of all poses of a round are one device
 * batch.  confs in place; tries / evals [B] optional. */
mi_status mi_cnn_refine_batch(mi_vina *, mi_scorer *, float *confs, int B, const mi_cnn_box *box, int max_iters,
                              float *energy, int32_t *tries, int32_t *evals);
/* Monte-Carlo with the CNN as the Metropolis energy: --cnn_scoring metrorescore / metrorefine (user_opts.h:27-33;
 * parallel_mc.cpp:145-155 runs monte_carlo with ig = the Vina grids and ig_metropolis = non_cache_cnn;
 * monte_carlo.cpp:44-47: update_energy = adjust_center + ig_metropolis->eval).  Same chain as mi_vina_mc_batch --
 * same mt19937 stream, mutation, BFGS on the Vina grids, container -- but every candidate's Metropolis energy is
 * non_cache_cnn::eval of what `model` holds: CNN loss of the scorer's ensemble + slope * out-of-box distances (search
 * box and the CNN cube re-centred on the pose, box->slope like non_cache's).  The chains of the call advance in lock
 * step; at each of the two update_energy points of a step ALL chains are scored in one CNN batch.  Stored energies
 * are CNN energies.  cnn_evals (optional) = CNN forward passes spent.  (--cnn_scoring all: mi_vina_mc_cnnall_batch.) */
mi_status mi_vina_mc_cnn_batch(mi_vina *, mi_scorer *, int B, const uint64_t *seeds, const float *corner1,
                               const float *corner2, const mi_mc_params *params, const mi_cnn_box *box, int32_t *out_n,
                               float *out_e, float *out_conf, float *out_coords, int32_t *evals, int32_t *cnn_evals);
/* --cnn_scoring all (parallel_mc.cpp:156-159): the same chain with non_cache_cnn as the igrid of the minimiser too --
 * quasi_newton inside the search minimises the CNN loss (+ penalties), monte_carlo.cpp:99-148 otherwise unchanged.  The
 * chain logic (mt19937 stream, mutate_conf, bfgs<> + fast_line_search, Metropolis, container) runs on the host for all B
 * chains in lock step; every round of evaluations -- eval_deriv inside the line searches, eval at the update_energy
 * points -- is one device batch (B CNN forwards [+ backwards] at once where the reference does one at B = 1).  Before a
 * chain's first update_energy its non_cache_cnn has no CNN cube (cnn_gd is default-constructed in the reference too).
 * No cache grids needed.  evals [B] = eval_deriv calls per chain; cnn_evals = CNN passes in total. */
mi_status mi_vina_mc_cnnall_batch(mi_vina *, mi_scorer *, int B, const uint64_t *seeds, const float *corner1,
                               const float *corner2, const mi_mc_params *params, const mi_cnn_box *box, int32_t *out_n,
                               float *out_e, float *out_conf, float *out_coords, int32_t *evals, int32_t *cnn_evals);
/* Latency probe for tools/bench_vina.py: device time (ms) of `reps` dependent evaluations per wave. */
mi_status mi_vina_eval_latency(mi_vina *, const float *confs, int B, int mode, int reps, float *ms_out);
void *mi_vina_stream(mi_vina *);

/* ---- typed-atom files ------------------------------------------------------------
 * `.gninatypes` = headerless array of struct { float x, y, z; int type; } (gninasrc/gninatyper/
 * gninatyper.cpp:30-36,65-75), type = smina type index 0..27: what gnina's `gninatyper` writes and
 * libmolgrid's providers read.  mi_read_gninatypes fills xyz [capacity][3] / smt [capacity] and *n_atoms
 * (call with xyz = smt = NULL to get the count).  Errors: MI_ERR_INVALID + mi_io_last_error().  Host only;
 * the C++ form is gnina_amd/host/typed_atoms.h. */
mi_status mi_read_gninatypes(const char *path, float *xyz, int32_t *smt, int capacity, int *n_atoms);
mi_status mi_write_gninatypes(const char *path, const float *xyz, const int32_t *smt, int n_atoms);
const char *mi_io_last_error(void);

/* ---- PDBQT files, no OpenBabel: gnina_amd/host/pdbqt.h restates parse_pdbqt.cpp
 * (ATOM columns, ROOT / BRANCH / ENDBRANCH / TORSDOF), postprocess_ligand (atom order, segment frames, mobility)
 * and model::initialize (distance-based bonds, adjust_smina_type, interacting pairs; model.cpp:560-720).
 * Receptor: rigid .pdbqt -> coordinates + smina types, ready for mi_scorer_set_receptor / mi_vina_set_receptor.
 * Ligand: handle whose mi_ligand_desc is what mi_vina_set_ligand takes; xyz = input coordinates in model order
 * (what mi_scorer_score_batch takes with desc.smt), serial = PDBQT atom numbers, conf0 [7+T] = the conformation
 * that reproduces the input pose.  is_text != 0: the first argument is the file's text.  Flexible residues
 * (BEGIN_RES) are not read.  Errors: NULL / MI_ERR_INVALID + mi_pdbqt_last_error() ("file:line: what"). */
typedef struct mi_pdbqt_ligand mi_pdbqt_ligand;
mi_status mi_pdbqt_read_receptor(const char *path, float *xyz, int32_t *smt, int capacity, int *n_atoms);
/* parse_receptor_pdbqt(rigid, flex) (parse_pdbqt.cpp:419-527): rigid receptor + flexible residues (BEGIN_RES ...
 * END_RES blocks with the ROOT / BRANCH grammar of a ligand).  Rows come back in DLScorer::setReceptor's order
 * (dl_scorer.cpp:93-193): n_movable movable side-chain atoms, n_inflex fixed atoms of the residues, then the rigid
 * atoms; types are those of the combined model (bonds never join a movable atom to the rigid part,
 * model.cpp:491-508).  Feed the rows to mi_scorer_set_receptor and declare rows 0 .. n_movable-1 with
 * mi_scorer_set_flex.  rigid / flex are paths, or the file contents when is_text != 0.  Same two-call protocol
 * (xyz = smt = NULL to query the size) and error reporting as mi_pdbqt_read_receptor. */
mi_status mi_pdbqt_read_receptor_flex(const char *rigid, const char *flex, int is_text, float *xyz, int32_t *smt,
                                      int capacity, int *n_atoms, int *n_movable, int *n_inflex);
/* gnina's `model` of a docking run WITH flexible residues: parse_receptor_pdbqt(rigid, flex) + m.append(ligand)
 * (molgetter.cpp:66-75,430-437; model::append model.cpp:176-226): the rigid atoms for the grids (rec_xyz / rec_smt,
 * [sizes[3]]) and ONE description of everything that moves -- atoms [flex movable | ligand | inflex], nodes [ligand |
 * residue trees], pairs with their kind -- ready for mi_vina_set_ligand.  conf0 [7 + T_ligand + T_flex] reproduces the
 * input; xyz [n_atoms][3] = input coordinates in model order.  sizes = {n_atoms, n_nodes, n_pairs, n_rigid,
 * n_flex_movable, n_inflex, T_ligand, T_flex}.  Arguments are paths, or file contents when is_text != 0.  The cache
 * needs grids for the types of ALL movable atoms (flexible side chains included). */
typedef struct mi_pdbqt_model mi_pdbqt_model;
mi_pdbqt_model *mi_pdbqt_model_open(const char *rigid, const char *flex, const char *ligand, int is_text);
void mi_pdbqt_model_close(mi_pdbqt_model *);
mi_status mi_pdbqt_model_sizes(const mi_pdbqt_model *, int32_t *sizes8);
mi_status mi_pdbqt_model_desc(const mi_pdbqt_model *, mi_ligand_desc *desc, const float **xyz, const float **conf0,
                              const float **rec_xyz, const int32_t **rec_smt, float *num_tors);
mi_pdbqt_ligand *mi_pdbqt_ligand_open(const char *path_or_text, int is_text);
void mi_pdbqt_ligand_close(mi_pdbqt_ligand *);
mi_status mi_pdbqt_ligand_sizes(const mi_pdbqt_ligand *, int *n_atoms, int *n_nodes, int *n_pairs, int *torsdof);
/* conf_independent_inputs::num_tors of the ligand (terms.cpp:39-106): the `num_tors` mi_vina_final_energies takes.
 * It counts rotatable bonds between heavy atoms that both have further heavy neighbours, not TORSDOF. */
mi_status mi_pdbqt_ligand_num_tors(const mi_pdbqt_ligand *, float *num_tors);
mi_status mi_pdbqt_ligand_desc(const mi_pdbqt_ligand *, mi_ligand_desc *desc, const float **xyz,
                               const int32_t **serial, const float **conf0);
/* One pose in gnina's .pdbqt output format (result_info::write, result_info.cpp:151-164; coordinates written back
 * into the input's own lines like context::writePDBQT, model.cpp:779-810): MODEL n / REMARK minimizedAffinity,
 * [minimizedRMSD if rmsd >= 0], [CNNscore if >= 0], [CNNaffinity if != 0] / ATOM lines / ENDMDL.  coords
 * [n_atoms][3] in model order.  Call with out = NULL to get the size (*needed, incl. the terminating 0). */
mi_status mi_pdbqt_write_pose(const mi_pdbqt_ligand *, const float *coords, int modelnum, float energy, float rmsd,
                              float cnnscore, float cnnaffinity, char *out, size_t capacity, size_t *needed);
/* One pose in gnina's .sdf output format (result_info::write's native SDF branch, result_info.cpp:117-160, around
 * sdfcontext::write's molecule block, model.cpp:827-907): V2000 counts / atoms (%10.4f) / bonds / M CHG / M ISO / M END,
 * then > <minimizedAffinity> (5 decimals), [> <minimizedRMSD> if rmsd >= 0], [> <CNNscore> if >= 0 (10 decimals)],
 * [> <CNNaffinity> and > <CNN_VS> = affinity * score if affinity != 0], [> <CNNaffinity_variance> if != 0], $$$$.
 * The connection table is the caller's (gnina keeps what OpenBabel read): elements [n_atoms][2] chars, atom_index
 * [n_atoms] (model atom of SDF atom i; NULL = identity), coords [.][3] in model order, bonds [n_bonds][3] = (a, b,
 * order) 0-based, props [n_props][3] = ('c' charge | 'i' isotope, atom, value).  out = NULL: size query. */
mi_status mi_sdf_write_pose(const char *name, int n_atoms, const char *elements, const int32_t *atom_index,
                            const float *coords, int n_bonds, const int32_t *bonds, int n_props, const int32_t *props,
                            float energy, float rmsd, float cnnscore, float cnnaffinity, float cnnvariance, char *out,
                            size_t capacity, size_t *needed);
const char *mi_pdbqt_last_error(void);

/* Per-kernel profiling for bench.py's roofline object: when enabled, every kernel launch of this
 * scorer is bracketed by HIP events on the scorer's stream.  mi_scorer_profile_json drains the
 * records and returns a JSON array [{kernel, launches, poses, ms_total, flops, bytes,
 * mfma_counted_launches, mfma_executed}], where flops / bytes are the ALGORITHMIC work of those
 * launches (2*MACs with unpadded channel counts; bytes as stated in DESIGN.md) and mfma_executed is
 * the number of 32x32x2 fp32 MFMA instructions (4,096 FLOPs each) the zero-skipping conv kernels
 * actually issued in mfma_counted_launches of them (device counters, this mode only; 0 / 0 for
 * kernels that execute every instruction).  The string stays valid until the next call on this scorer. */
mi_status mi_scorer_enable_profile(mi_scorer *, int on);
const char *mi_scorer_profile_json(mi_scorer *);

#ifdef __cplusplus
}
#endif
#endif /* MI_GNINA_H_ */
