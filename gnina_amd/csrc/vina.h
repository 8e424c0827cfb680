// vina.h -- device-side data model of the smina/Vina scoring + BFGS engine (vina.hip, vina_host.cpp).
//
// Reference rows: a11 precalculate_linear (precalculate.h:165-272), a12 cache::populate
// (cache.cpp:104-184), a13 cache::eval_deriv -> grid::evaluate_aux (cache.cpp:65-83, grid.cpp:96-186),
// a14 eval_interacting_pairs_deriv (model.cu:38-60), a15 model::set / tree (tree.h), a16 bfgs<>
// (bfgs.h:357-502).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mig {

constexpr int kVinaTypes = 28;

struct VinaGridGeom {
  float init[3], factor[3], factor_inv[3], dim_m1[3];
  int dim[3];
};

struct VinaEnv {
  // pair tables
  const float2 *smooth;  // [npairs][n] (E, dor)
  const float *fast;     // [npairs][n]
  int n;
  float factor, cutoff_sqr;
  // receptor cache grids
  VinaGridGeom geom;
  const float *grid_data;       // all type grids back to back
  long grid_off[kVinaTypes];    // float offset of type t's grid, -1 if absent
  float slope;
  // direct (grid-free) receptor term = the non_cache igrid (non_cache.cpp:52-83,125-179), and the
  // exact pair functions = precalculate_exact (precalculate.h:452-494)
  int direct, exact;
  int accurate_ls;  // minimization_params::BFGSAccurateLineSearch (--accurate_line_search) instead of fast_line_search
  // precalculate_splines (precalculate.h:277-449, splines.h): per type pair a clamped cubic spline of E(r) over
  // [0, cutoff] in sp_n intervals, (a, b, c, d) per interval; null = the linear tables above
  const float4 *spline;
  int sp_n;
  float sp_fraction, cutoff;
  int stage;  // copy the ligand description into LDS (set by the launchers)
  int strict;  // energy sums in the reference's order instead of the butterfly (mi_vina_set_strict_order; vina.hip seq_add)
  const float4 *rec;  // (x, y, z, smt bits)
  int n_rec;
  float w5[5];
  float box_begin[3], box_end[3];
  // --user_grid (main.cpp:1342-1350): a second trilinear grid with its own dims, added per atom in
  // non_cache(_cnn)::eval_deriv (non_cache.cpp:168-173, non_cache_cnn.cpp:141-150); null = none
  VinaGridGeom ug_geom;
  const float *ug_data;
  int ug_model;  // MODE 2 adds model::eval's user-grid sum (MI_VINA_USER_TERM)
  // screen mode of the batch kernels (mi_vina_*_screen): item b is a conformation of ligand item_lig[b] of `ligs`;
  // rows of confs / change / coords are then strided by the maxima over the set
  const struct VinaLigand *ligs;
  const int *item_lig;
  const int *lig_iters;  // refine: per-ligand BFGS iteration cap
  int conf_stride, change_stride, coord_stride;
};

struct VinaLigand {
  int n_atoms, n_nodes, n_pairs;
  const int *smt;            // [n_atoms]
  const float *local_xyz;    // [n_atoms][3]
  const int *node_of_atom;   // [n_atoms]
  const int *parent;         // [n_nodes]
  const int *abeg, *aend;    // [n_nodes]
  const float *rel_origin;   // [n_nodes][3]
  const float *rel_axis;     // [n_nodes][3]
  const int *child_start;    // [n_nodes+1] CSR of children in increasing index order
  const int *child_list;
  const int2 *pairs;         // [n_pairs] (a, b), a < b
  // per-atom lists of pair-force contributions: atom i owns slots [slot_start[i], slot_start[i+1]) (both
  // multiples of 4; the tail of a list is zero padding), its pairs in pair order; pair p writes -f to
  // pair_slots[p].x (atom a's list) and +f to pair_slots[p].y (atom b's list)
  const int *slot_start;       // [n_atoms+1]
  const int2 *pair_slots;      // [n_pairs]
  int n_heavy;
  const int *heavy_list;       // [n_heavy] indices of the non-hydrogen atoms
  const int *depth;            // [n_nodes] tree level of every node (root 0)
  int n_levels;                // 1 + the deepest level
  // Flexible receptor residues (model.h: atoms = [flex movable | ligand | inflex]; tree.h:266-283,374-393):
  //  parent[k] == -2: node k is a residue's first_segment -- its frame hangs off the world (rel_origin / rel_axis are
  //  absolute); node_of_atom[i] == -1: an inflex atom, fixed at local_xyz, a partner in pairs only;
  //  atoms [0, n_movable) get the receptor term; pair_cap[p] != 0: a pair of model::other_pairs (curl cap v[2])
  //  instead of the ligand's own (v[0]); [lig_begin, lig_end) = the ligand's atoms (gyration radius).
  int n_movable;
  const int *pair_cap;         // [n_pairs] or nullptr (plain ligand)
  int lig_begin, lig_end;
};

struct VinaMcArgs {
  int n_steps, max_iters, num_saved;
  float temperature, amplitude, min_rmsd;
  float hunt[3], auth[3];
  float c1[3], c2[3];            // search box corners
  const unsigned long long *seeds;  // [B]
  unsigned *mt;  // [B][mt_team][624] MT19937 states, seeded on the host (one private copy per wave of a chain's team)
  int mt_team;   // copies per chain (>= the team size the launch ends up with)
  float *scratch_e, *scratch_conf, *scratch_coords;  // per-chain physical container
  float *out_e, *out_conf, *out_coords;              // sorted output [B][num_saved]...
  int *out_n, *evals;
  long long *prof;  // optional [B][12] phase timing, see vina_mc_kernel
  // screen mode (mi_vina_mc_screen): chain b docks ligand chain_lig[b] of `ligs` with its own step / iteration
  // counts; containers use the common strides below (floats per saved conformation / coordinate set)
  const VinaLigand *ligs;
  const int *chain_lig, *lig_steps, *lig_iters;
  const int *order;  // optional [B]: workgroup g runs chain order[g] (longest searches first)
  int conf_stride, coord_stride;
};

// resumable chains of vina_mc_cnn_kernel (CNN as the Metropolis energy)
struct VinaMcCnnState {
  int phase, step;     // see vina_mc_cnn_kernel
  float *st_f;         // [B][f_stride]
  int *st_i;           // [B][i_stride]
  int f_stride, i_stride;
  const float *ext_e;  // [B] non_cache_cnn::eval of what model_out held
  float *model_out;    // [B][7 + T]
};

struct VinaExtArgs {  // non_cache_cnn: externally computed receptor term (CNN loss + per-atom gradient)
  const float *forces;      // [B][n_atoms][3] or nullptr (energy only)
  const float *e_in;        // [B] or nullptr
  int use_box;              // search box gd active
  float box_begin[3], box_end[3];
  const float *cnn_center;  // [B][3] centre of the CNN cube cnn_gd, or nullptr (cube inactive)
  float cnn_half;           // half its side
  float slope;
  // cnn_options::mix_emp_force / mix_emp_energy / empirical_weight (user_opts.h:46-51); v = curl cap
  int mix_force, mix_energy;
  float weight, v;
  // model::add_minus_forces (model.cu:247-259) hands the scorer's gradient -- one entry per movable atom, hydrogens
  // included (cnn_torch_scorer.cpp:208-228) -- to the non-hydrogen atoms with a counter that only advances on
  // non-hydrogen atoms: the k-th heavy atom receives entry k, not its own.  0 = exactly that (the reference's
  // behaviour: what non_cache_cnn::eval_deriv sees in m.minus_forces); 1 = every atom its own gradient.
  int per_atom_forces;
};

struct VinaPopulateArgs {
  const float4 *rec;  // (x, y, z, smt as float bits)
  int n_rec;
  const float *fast;
  int n;
  float factor, cutoff_sqr;
  VinaGridGeom geom;
  int lig_type;
  float *out;  // [(dimz)][(dimy)][(dimx)], x fastest
  // per dimension and lattice index: the candidate brick of the point's 3 A cell (szv_grid_cache::get), see
  // mi_vina_build_cache.  [dimx + dimy + dimz] (lo, hi)
  const float2 *brick;
  const float4 *spline;  // precalculate_splines instead of `fast` (see VinaEnv), or null
  int sp_n;
  float sp_fraction, cutoff;
  // --user_grid: added to every point (cache.cpp:177-179); null = none
  VinaGridGeom ug_geom;
  const float *ug_data;
  float ug_slope;
};

size_t vina_wave_lds_bytes(int n_atoms, int n_nodes, int n_pairs, bool bfgs, bool stage);
void launch_vina_populate(const VinaPopulateArgs &a, hipStream_t s);
void launch_vina_sincos_probe(const float *x, int n, float *sn, float *cs, hipStream_t s);
void launch_vina_explog_probe(const float *x, int n, float *ex, float *lg, hipStream_t s);
void launch_vina_acos_probe(const float *x, int n, float *ac, hipStream_t s);
// confs [B][7+T]; energy [B]; change [B][6+T] or null; coords [B][n_atoms][3] or null
void launch_vina_eval(const VinaEnv &env, const VinaLigand &lig, const float *confs, int B, float v0, float v1,
                      float v2, int with_deriv, float *energy, float *change, float *coords, hipStream_t s);
// model::set(conf) only: coords [B][n_atoms][3]
void launch_vina_coords(const VinaEnv &env, const VinaLigand &lig, const float *confs, int B, float *coords,
                        hipStream_t s);
// energy [B] = e_in + box penalties; change [B][6+T] (optional) from the external forces + penalty forces
void launch_vina_extforce(const VinaEnv &env, const VinaLigand &lig, const float *confs, int B, const VinaExtArgs &a,
                          float *energy, float *change, hipStream_t s);
// in-place BFGS (quasi_newton, bfgs.h:357-502 with fast_line_search); evals [B] optional
size_t vina_mc_lds_bytes(int n_atoms, int n_nodes, int n_pairs, int n_heavy, int num_saved, bool stage, int waves_per_chain);
void launch_vina_cache_coords(const VinaEnv &env, const float *coords, const int *smt, int n_atoms, int B, float v,
                              float *energy, float *minus_forces, hipStream_t s);
void launch_vina_mc_cnn(const VinaEnv &env, const VinaLigand &lig, const VinaMcArgs &a, const VinaMcCnnState &st, int B,
                        hipStream_t s);
int vina_mc_team(int B);  // waves per chain the Monte-Carlo kernel uses for B chains
// `lig` sizes the LDS workspace (screen mode: counts = the maxima over the set, pointers unused)
void launch_vina_mc(const VinaEnv &env, const VinaLigand &lig, const VinaMcArgs &a, int B, hipStream_t s);
void launch_vina_bfgs(const VinaEnv &env, const VinaLigand &lig, float *confs, int B, float v0, float v1, float v2,
                      int max_iters, float *energy, float *grad, int *evals, hipStream_t s);
// refine_structure (main.cpp:131-171): BFGS on the direct receptor term with the slope ladder 10, 100, ...
void launch_vina_eval_repeat(const VinaEnv &env, const VinaLigand &lig, const float *confs, int B, int mode, int reps,
                             float *energy, hipStream_t s);
void launch_vina_refine(const VinaEnv &env, const VinaLigand &lig, float *confs, int B, float v0, float v1, float v2,
                        int max_iters, float *energy, int *tries, hipStream_t s);

}  // namespace mig
