"""ctypes binding of libmi_gnina.so (include/mi_gnina.h) -- the parity-test / bench harness side
of the C ABI.  There is NO fallback: if the HIP library is missing this module raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 2  # include/mi_gnina.h MI_GNINA_ABI_VERSION: the struct layouts this binding was written against
LIB_PATH = os.environ.get("MI_GNINA_LIB", os.path.join(_HERE, "lib", "libmi_gnina.so"))

MI_OK = 0
MI_LIG_ON_DEVICE = 1
MI_OUT_ON_DEVICE = 2
MI_CENTER_TYPED_ONLY = 4

# every function include/mi_gnina.h declares (tests check the library exports all of them)
SYMBOLS = [
    "mi_gnina_init", "mi_gnina_device_count", "mi_gnina_abi_version", "mi_last_error",
    "mi_model_load", "mi_model_load_file", "mi_model_load_file_ex", "mi_model_retain", "mi_model_release", "mi_model_info",
    "mi_model_name", "mi_model_type_channel",
    "mi_scorer_create", "mi_scorer_destroy", "mi_scorer_num_models", "mi_scorer_set_receptor",
    "mi_scorer_score_batch", "mi_scorer_score_batch_ex", "mi_scorer_last_model_outputs",
    "mi_voxelize_batch", "mi_model_forward_grids", "mi_scorer_score_grad", "mi_scorer_score_ragged", "mi_read_gninatypes", "mi_pdbqt_read_receptor", "mi_pdbqt_read_receptor_flex", "mi_pdbqt_model_open", "mi_pdbqt_model_close", "mi_pdbqt_model_sizes", "mi_pdbqt_model_desc", "mi_pdbqt_ligand_open", "mi_pdbqt_ligand_close", "mi_pdbqt_ligand_sizes", "mi_pdbqt_ligand_num_tors", "mi_pdbqt_ligand_desc", "mi_pdbqt_write_pose", "mi_sdf_write_pose", "mi_pdbqt_last_error", "mi_write_gninatypes", "mi_io_last_error", "mi_scorer_set_precision", "mi_model_supports_gradient", "mi_vina_coords_batch", "mi_vina_cache_eval_coords", "mi_cnn_eval_batch", "mi_vina_mc_cnn_batch", "mi_vina_mc_cnnall_batch", "mi_cnn_refine_batch", "mi_scorer_set_flex", "mi_scorer_set_rotations", "mi_scorer_score_flex", "mi_scorer_stream", "mi_scorer_synchronize", "mi_scorer_h2_fallbacks",
    "mi_scorer_set_chunk", "mi_scorer_enable_timing", "mi_scorer_last_timing",
    "mi_scorer_enable_profile", "mi_scorer_profile_json",
    "mi_vina_create", "mi_vina_destroy", "mi_vina_table_size", "mi_vina_table", "mi_vina_set_receptor",
    "mi_vina_build_cache", "mi_vina_cache_grid", "mi_user_grid_parse", "mi_vina_set_user_grid", "mi_vina_set_approximation", "mi_vina_pair_eval", "mi_vina_set_line_search", "mi_vina_set_strict_order", "mi_debug_sincos", "mi_debug_explog", "mi_debug_acos", "mi_debug_split_f16", "mi_debug_h2_layout", "mi_debug_read_activation", "mi_debug_read_candidates", "mi_debug_vox_stress", "mi_gnina_set_option", "mi_gnina_options", "mi_scorer_flex_count", "mi_pool_create", "mi_pool_destroy", "mi_pool_size", "mi_pool_set_receptor", "mi_pool_score_batch", "mi_pool_score_ragged", "mi_pool_info_json", "mi_vina_set_ligand", "mi_vina_eval_batch",
    "mi_vina_bfgs_batch", "mi_vina_stream", "mi_vina_mc_batch", "mi_vina_ligand_heavy_atoms",
    "mi_vina_set_screen", "mi_vina_screen_size", "mi_vina_screen_dims", "mi_vina_mc_screen",
    "mi_vina_eval_screen", "mi_vina_refine_screen", "mi_vina_final_energies_screen",
    "mi_vina_refine_batch", "mi_vina_final_energies", "mi_rank_poses", "mi_merge_mc_outputs", "mi_vina_eval_latency",
    "mi_vina_pool_create", "mi_vina_pool_destroy", "mi_vina_pool_size", "mi_vina_pool_configure", "mi_vina_pool_mc_batch",
    "mi_vina_pool_mc_screen", "mi_vina_pool_info_json",
]

_lib = None


class LigandDesc(C.Structure):
    """mi_ligand_desc (include/mi_gnina.h)"""
    _fields_ = [("n_atoms", C.c_int32), ("smt", C.c_void_p), ("local_xyz", C.c_void_p), ("n_nodes", C.c_int32),
                ("node_parent", C.c_void_p), ("node_atom_begin", C.c_void_p), ("node_atom_end", C.c_void_p),
                ("node_rel_origin", C.c_void_p), ("node_rel_axis", C.c_void_p), ("n_pairs", C.c_int32),
                ("pairs", C.c_void_p),
                # flexible residues: zero / NULL for a plain ligand
                ("n_movable", C.c_int32), ("pair_kind", C.c_void_p), ("lig_begin", C.c_int32), ("lig_end", C.c_int32)]


class CnnBox(C.Structure):
    """mi_cnn_box (include/mi_gnina.h): the two out-of-box penalty regions of non_cache_cnn"""
    _fields_ = [("use_search_box", C.c_int32), ("box_begin", C.c_float * 3), ("box_end", C.c_float * 3),
                ("cnn_dimension", C.c_float), ("slope", C.c_float), ("mix_emp_force", C.c_int32),
                ("mix_emp_energy", C.c_int32), ("empirical_weight", C.c_float), ("v", C.c_float),
                ("per_atom_forces", C.c_int32)]

    @classmethod
    def make(cls, cnn_dimension, box_begin=None, box_end=None, slope=10.0, mix_emp_force=False,
             mix_emp_energy=False, empirical_weight=1.0, v=1000.0, per_atom_forces=False):
        """per_atom_forces False (default) = the reference's model::add_minus_forces indexing, see mi_gnina.h"""
        use = box_begin is not None
        bb = (C.c_float * 3)(*(box_begin if use else (0, 0, 0)))
        be = (C.c_float * 3)(*(box_end if use else (0, 0, 0)))
        return cls(1 if use else 0, bb, be, cnn_dimension, slope, int(mix_emp_force), int(mix_emp_energy),
                   empirical_weight, v, int(per_atom_forces))


class McParams(C.Structure):
    """mi_mc_params (include/mi_gnina.h); defaults = gnina's (monte_carlo.h:38-40, main.cpp:441-463)"""
    _fields_ = [("n_steps", C.c_int32), ("max_iters", C.c_int32), ("num_saved", C.c_int32),
                ("temperature", C.c_float), ("mutation_amplitude", C.c_float), ("min_rmsd", C.c_float),
                ("hunt_cap", C.c_float * 3), ("authentic_v", C.c_float * 3)]

    @classmethod
    def default(cls, n_steps, max_iters, num_saved=50):
        return cls(n_steps, max_iters, num_saved, 1.2, 2.0, 1.0, (C.c_float * 3)(10, 10, 10),
                   (C.c_float * 3)(1000, 1000, 1000))


class MiGninaError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiGninaError(
                f"{LIB_PATH} not built: run `python __graft_entry__.py build` (hipcc, gfx950). "
                "gnina_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        f32p, i32p, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
        L.mi_gnina_init.argtypes = [C.c_int]
        L.mi_gnina_init.restype = C.c_int
        L.mi_gnina_device_count.restype = C.c_int
        L.mi_gnina_abi_version.restype = C.c_int
        if L.mi_gnina_abi_version() != ABI_VERSION:
            raise MiGninaError(f"{LIB_PATH} has ABI version {L.mi_gnina_abi_version()}, this binding expects {ABI_VERSION} "
                               "(struct layouts differ): rebuild with `python __graft_entry__.py build`")
        L.mi_last_error.restype = C.c_char_p
        L.mi_gnina_set_option.argtypes = [C.c_char_p, C.c_char_p]
        L.mi_gnina_set_option.restype = C.c_int
        L.mi_gnina_options.restype = C.c_char_p
        L.mi_model_load.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
        L.mi_model_load.restype = vp
        L.mi_model_load_file.argtypes = [C.c_char_p]
        L.mi_model_load_file.restype = vp
        L.mi_model_load_file_ex.argtypes = [C.c_char_p, C.c_float, C.c_float]
        L.mi_model_load_file_ex.restype = vp
        L.mi_model_retain.argtypes = [vp]
        L.mi_model_retain.restype = None
        L.mi_model_release.argtypes = [vp]
        L.mi_model_release.restype = None
        L.mi_model_info.argtypes = [vp, f32p, f32p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mi_model_info.restype = C.c_int
        L.mi_model_name.argtypes = [vp]
        L.mi_model_name.restype = C.c_char_p
        L.mi_model_type_channel.argtypes = [vp, C.c_int, C.c_int, f32p]
        L.mi_model_type_channel.restype = C.c_int
        L.mi_scorer_create.argtypes = [C.POINTER(vp), C.c_int]
        L.mi_scorer_create.restype = vp
        L.mi_scorer_destroy.argtypes = [vp]
        L.mi_scorer_destroy.restype = None
        L.mi_scorer_num_models.argtypes = [vp]
        L.mi_scorer_num_models.restype = C.c_int
        L.mi_scorer_set_receptor.argtypes = [vp, vp, vp, C.c_int]
        L.mi_scorer_set_receptor.restype = C.c_int
        L.mi_scorer_score_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.mi_scorer_score_batch.restype = C.c_int
        L.mi_scorer_score_batch_ex.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_uint]
        L.mi_scorer_score_batch_ex.restype = C.c_int
        L.mi_scorer_score_grad.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
        L.mi_scorer_score_grad.restype = C.c_int
        L.mi_scorer_set_flex.argtypes = [vp, vp, C.c_int]
        L.mi_scorer_set_rotations.argtypes = [vp, vp, C.c_int]
        L.mi_read_gninatypes.argtypes = [C.c_char_p, vp, vp, C.c_int, C.POINTER(C.c_int)]
        L.mi_write_gninatypes.argtypes = [C.c_char_p, vp, vp, C.c_int]
        L.mi_io_last_error.restype = C.c_char_p
        L.mi_pdbqt_read_receptor.argtypes = [C.c_char_p, vp, vp, C.c_int, C.POINTER(C.c_int)]
        L.mi_pdbqt_read_receptor_flex.argtypes = [C.c_char_p, C.c_char_p, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int),
                                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mi_pdbqt_read_receptor_flex.restype = C.c_int
        L.mi_pdbqt_model_open.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.mi_pdbqt_model_open.restype = vp
        L.mi_pdbqt_model_close.argtypes = [vp]
        L.mi_pdbqt_model_close.restype = None
        L.mi_pdbqt_model_sizes.argtypes = [vp, vp]
        L.mi_pdbqt_model_desc.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.mi_pdbqt_ligand_open.argtypes = [C.c_char_p, C.c_int]
        L.mi_pdbqt_ligand_open.restype = vp
        L.mi_pdbqt_ligand_close.argtypes = [vp]
        L.mi_pdbqt_ligand_close.restype = None
        L.mi_pdbqt_ligand_sizes.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
        L.mi_pdbqt_ligand_num_tors.argtypes = [vp, C.POINTER(C.c_float)]
        L.mi_pdbqt_ligand_desc.argtypes = [vp, vp, vp, vp, vp]
        L.mi_pdbqt_last_error.restype = C.c_char_p
        L.mi_pdbqt_write_pose.argtypes = [vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
        L.mi_scorer_set_precision.argtypes = [vp, C.c_int]
        L.mi_scorer_score_ragged.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.mi_vina_coords_batch.argtypes = [vp, vp, C.c_int, vp]
        L.mi_vina_cache_eval_coords.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]
        L.mi_cnn_eval_batch.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp]
        L.mi_vina_mc_cnn_batch.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.mi_vina_mc_cnnall_batch.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.mi_vina_mc_cnnall_batch.restype = C.c_int
        L.mi_cnn_refine_batch.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
        L.mi_scorer_score_flex.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
        L.mi_model_supports_gradient.argtypes = [vp]
        L.mi_model_supports_gradient.restype = C.c_int
        L.mi_scorer_last_model_outputs.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
        L.mi_scorer_last_model_outputs.restype = C.c_int
        L.mi_voxelize_batch.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_uint]
        L.mi_voxelize_batch.restype = C.c_int
        L.mi_model_forward_grids.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp]
        L.mi_model_forward_grids.restype = C.c_int
        L.mi_scorer_stream.argtypes = [vp]
        L.mi_scorer_stream.restype = vp
        L.mi_scorer_synchronize.argtypes = [vp]
        L.mi_scorer_synchronize.restype = C.c_int
        L.mi_scorer_h2_fallbacks.argtypes = [vp]
        L.mi_scorer_h2_fallbacks.restype = C.c_int
        L.mi_scorer_set_chunk.argtypes = [vp, C.c_int]
        L.mi_scorer_set_chunk.restype = C.c_int
        L.mi_scorer_enable_timing.argtypes = [vp, C.c_int]
        L.mi_scorer_enable_timing.restype = C.c_int
        L.mi_scorer_last_timing.argtypes = [vp, f32p]
        L.mi_scorer_last_timing.restype = C.c_int
        L.mi_scorer_enable_profile.argtypes = [vp, C.c_int]
        L.mi_scorer_enable_profile.restype = C.c_int
        L.mi_scorer_profile_json.argtypes = [vp]
        L.mi_scorer_profile_json.restype = C.c_char_p
        L.mi_vina_create.argtypes = [vp, C.c_float, C.c_float]
        L.mi_vina_create.restype = vp
        L.mi_vina_destroy.argtypes = [vp]
        L.mi_vina_destroy.restype = None
        L.mi_vina_table_size.argtypes = [vp]
        L.mi_vina_table_size.restype = C.c_int
        L.mi_vina_table.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
        L.mi_vina_table.restype = C.c_int
        L.mi_vina_set_receptor.argtypes = [vp, vp, vp, C.c_int]
        L.mi_vina_set_receptor.restype = C.c_int
        L.mi_vina_build_cache.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_float]
        L.mi_vina_build_cache.restype = C.c_int
        L.mi_vina_cache_grid.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.mi_vina_cache_grid.restype = C.c_int
        L.mi_user_grid_parse.argtypes = [C.c_char_p, C.c_size_t, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mi_user_grid_parse.restype = C.c_int
        L.mi_vina_set_user_grid.argtypes = [vp, vp, vp, vp, vp, C.c_float]
        L.mi_vina_set_user_grid.restype = C.c_int
        L.mi_vina_set_approximation.argtypes = [vp, C.c_int, C.c_float]
        L.mi_vina_set_approximation.restype = C.c_int
        L.mi_vina_pair_eval.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
        L.mi_vina_pair_eval.restype = C.c_int
        L.mi_vina_set_line_search.argtypes = [vp, C.c_int]
        L.mi_vina_set_line_search.restype = C.c_int
        L.mi_vina_set_strict_order.argtypes = [vp, C.c_int]
        L.mi_vina_set_strict_order.restype = C.c_int
        L.mi_debug_sincos.argtypes = [f32p, C.c_int, f32p, f32p]
        L.mi_debug_sincos.restype = C.c_int
        L.mi_debug_explog.argtypes = [f32p, C.c_int, f32p, f32p]
        L.mi_debug_explog.restype = C.c_int
        L.mi_debug_acos.argtypes = [f32p, C.c_int, f32p]
        L.mi_debug_acos.restype = C.c_int
        L.mi_debug_split_f16.argtypes = [f32p, C.c_int, C.c_float, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), f32p]
        L.mi_debug_split_f16.restype = C.c_int
        L.mi_debug_read_activation.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i32p, f32p, C.c_size_t]
        L.mi_debug_read_activation.restype = C.c_int
        L.mi_scorer_flex_count.argtypes = [vp]
        L.mi_scorer_flex_count.restype = C.c_int
        L.mi_pool_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_char_p), C.c_int]
        L.mi_pool_create.restype = vp
        L.mi_pool_destroy.argtypes = [vp]
        L.mi_pool_destroy.restype = None
        L.mi_pool_size.argtypes = [vp]
        L.mi_pool_size.restype = C.c_int
        L.mi_pool_set_receptor.argtypes = [vp, vp, vp, C.c_int]
        L.mi_pool_set_receptor.restype = C.c_int
        L.mi_pool_score_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_uint]
        L.mi_pool_score_batch.restype = C.c_int
        L.mi_pool_score_ragged.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.mi_pool_score_ragged.restype = C.c_int
        L.mi_pool_info_json.argtypes = [vp]
        L.mi_pool_info_json.restype = C.c_char_p
        L.mi_vina_set_ligand.argtypes = [vp, C.POINTER(LigandDesc)]
        L.mi_vina_set_ligand.restype = C.c_int
        L.mi_vina_eval_batch.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
        L.mi_vina_eval_batch.restype = C.c_int
        L.mi_vina_bfgs_batch.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
        L.mi_vina_bfgs_batch.restype = C.c_int
        L.mi_vina_mc_batch.argtypes = [vp, C.c_int, vp, vp, vp, C.POINTER(McParams), vp, vp, vp, vp, vp]
        L.mi_vina_mc_batch.restype = C.c_int
        L.mi_vina_set_screen.argtypes = [vp, C.c_int, vp]
        L.mi_vina_set_screen.restype = C.c_int
        L.mi_vina_screen_size.argtypes = [vp]
        L.mi_vina_screen_size.restype = C.c_int
        L.mi_vina_screen_dims.argtypes = [vp, i32p, i32p, i32p]
        L.mi_vina_screen_dims.restype = C.c_int
        L.mi_vina_mc_screen.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.mi_vina_mc_screen.restype = C.c_int
        L.mi_vina_eval_screen.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
        L.mi_vina_eval_screen.restype = C.c_int
        L.mi_vina_refine_screen.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp]
        L.mi_vina_refine_screen.restype = C.c_int
        L.mi_vina_final_energies_screen.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp]
        L.mi_vina_final_energies_screen.restype = C.c_int
        L.mi_vina_ligand_heavy_atoms.argtypes = [vp]
        L.mi_vina_ligand_heavy_atoms.restype = C.c_int
        L.mi_vina_refine_batch.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp]
        L.mi_vina_refine_batch.restype = C.c_int
        L.mi_vina_final_energies.argtypes = [vp, vp, C.c_int, vp, C.c_float, vp, vp]
        L.mi_vina_final_energies.restype = C.c_int
        L.mi_rank_poses.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
        L.mi_rank_poses.restype = C.c_int
        L.mi_merge_mc_outputs.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp,
                                          vp, vp]
        L.mi_merge_mc_outputs.restype = C.c_int
        L.mi_vina_eval_latency.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
        L.mi_vina_eval_latency.restype = C.c_int
        L.mi_vina_stream.argtypes = [vp]
        L.mi_vina_stream.restype = vp
        _lib = L
    return _lib


def check(status):
    if status != MI_OK:
        raise MiGninaError(f"mi_gnina error {status}: {lib().mi_last_error().decode()}")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


WEIGHTS_DIR = os.path.join(_HERE, "weights")


def read_gninatypes(path):
    """`.gninatypes` (gninatyper.cpp:30-36) -> (xyz [n,3] float32, smt [n] int32)"""
    n = C.c_int()
    if lib().mi_read_gninatypes(path.encode(), None, None, 0, C.byref(n)) != MI_OK:
        raise MiGninaError(lib().mi_io_last_error().decode())
    xyz, smt = np.empty((n.value, 3), dtype=np.float32), np.empty(n.value, dtype=np.int32)
    if lib().mi_read_gninatypes(path.encode(), _ptr(xyz), _ptr(smt), n.value, C.byref(n)) != MI_OK:
        raise MiGninaError(lib().mi_io_last_error().decode())
    return xyz, smt


def write_gninatypes(path, xyz, smt):
    xyz, smt = _f32(xyz).reshape(-1, 3), _i32(smt)
    if lib().mi_write_gninatypes(path.encode(), _ptr(xyz), _ptr(smt), len(smt)) != MI_OK:
        raise MiGninaError(lib().mi_io_last_error().decode())


def read_pdbqt_receptor(path):
    """rigid receptor .pdbqt -> (xyz [n,3], smt [n]) with gnina's typing (parse_pdbqt_rigid + model::initialize)"""
    n = C.c_int()
    if lib().mi_pdbqt_read_receptor(path.encode(), None, None, 0, C.byref(n)) != MI_OK:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    xyz, smt = np.empty((n.value, 3), dtype=np.float32), np.empty(n.value, dtype=np.int32)
    if lib().mi_pdbqt_read_receptor(path.encode(), _ptr(xyz), _ptr(smt), n.value, C.byref(n)) != MI_OK:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    return xyz, smt


def read_pdbqt_receptor_flex(rigid, flex, is_text=False):
    """rigid receptor + flexible residues (.pdbqt paths, or contents with is_text) -> (xyz [n,3], smt [n], n_movable,
    n_inflex); rows: movable side-chain atoms, the residues' fixed atoms, the rigid part (DLScorer::setReceptor's
    order: declare rows 0 .. n_movable-1 with Scorer.set_flex)"""
    n, nm, ni = C.c_int(), C.c_int(), C.c_int()
    a, b = rigid.encode(), flex.encode()
    if lib().mi_pdbqt_read_receptor_flex(a, b, int(is_text), None, None, 0, C.byref(n), C.byref(nm), C.byref(ni)) != MI_OK:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    xyz, smt = np.empty((n.value, 3), dtype=np.float32), np.empty(n.value, dtype=np.int32)
    if lib().mi_pdbqt_read_receptor_flex(a, b, int(is_text), _ptr(xyz), _ptr(smt), n.value, C.byref(n), C.byref(nm),
                                         C.byref(ni)) != MI_OK:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    return xyz, smt, nm.value, ni.value


def read_pdbqt_ligand(path_or_text, is_text=False):
    """ligand .pdbqt -> dict in the layout of gnina_amd.synth.make_ligand_tree (what Vina.set_ligand takes) plus
    coords0 / conf0 / serial / torsdof"""
    h = lib().mi_pdbqt_ligand_open(path_or_text.encode(), 1 if is_text else 0)
    if not h:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    try:
        d = LigandDesc()
        pxyz, pser, pconf = C.POINTER(C.c_float)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
        na, nn, npairs, tors = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().mi_pdbqt_ligand_sizes(h, C.byref(na), C.byref(nn), C.byref(npairs), C.byref(tors)))
        check(lib().mi_pdbqt_ligand_desc(h, C.byref(d), C.byref(pxyz), C.byref(pser), C.byref(pconf)))
        na, nn, npairs = na.value, nn.value, npairs.value

        def arr(p, shape, dt):
            n = int(np.prod(shape))
            if n == 0:
                return np.zeros(shape, dtype=dt)
            ct = C.c_float if dt == np.float32 else C.c_int32
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,)).reshape(shape).astype(dt, copy=True)

        return {"smt": arr(d.smt, (na,), np.int32), "local_xyz": arr(d.local_xyz, (na, 3), np.float32),
                "parent": arr(d.node_parent, (nn,), np.int32), "abeg": arr(d.node_atom_begin, (nn,), np.int32),
                "aend": arr(d.node_atom_end, (nn,), np.int32), "rel_origin": arr(d.node_rel_origin, (nn, 3), np.float32),
                "rel_axis": arr(d.node_rel_axis, (nn, 3), np.float32), "pairs": arr(d.pairs, (npairs, 2), np.int32),
                "coords0": arr(pxyz, (na, 3), np.float32), "serial": arr(pser, (na,), np.int32),
                "conf0": arr(pconf, (7 + nn - 1,), np.float32), "n_tors": nn - 1, "torsdof": tors.value,
                "num_tors": _ligand_num_tors(h)}
    finally:
        lib().mi_pdbqt_ligand_close(h)


def read_pdbqt_model(rigid, flex, ligand, is_text=False):
    """rigid receptor + flexible residues + ligand (.pdbqt) -> (rec_xyz, rec_smt, desc): desc is a dict in the layout
    of read_pdbqt_ligand plus n_movable / pair_kind / lig_begin / lig_end -- atoms [flex movable | ligand | inflex],
    conf0 [7 + T_ligand + T_flex] -- what Vina.set_ligand takes for a search with flexible side chains."""
    h = lib().mi_pdbqt_model_open(rigid.encode(), flex.encode(), ligand.encode(), 1 if is_text else 0)
    if not h:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    try:
        sz = np.zeros(8, dtype=np.int32)
        check(lib().mi_pdbqt_model_sizes(h, _ptr(sz)))
        na, nn, npairs, nrig, nfm, ninf, tl, tf = (int(x) for x in sz)
        d = LigandDesc()
        pxyz, pconf, prx = (C.POINTER(C.c_float)() for _ in range(3))
        prs = C.POINTER(C.c_int32)()
        nt = C.c_float()
        check(lib().mi_pdbqt_model_desc(h, C.byref(d), C.byref(pxyz), C.byref(pconf), C.byref(prx), C.byref(prs), C.byref(nt)))

        def arr(p, shape, dt):
            n = int(np.prod(shape))
            if n == 0:
                return np.zeros(shape, dtype=dt)
            ct = C.c_float if dt == np.float32 else C.c_int32
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,)).reshape(shape).astype(dt, copy=True)

        desc = {"smt": arr(d.smt, (na,), np.int32), "local_xyz": arr(d.local_xyz, (na, 3), np.float32),
                "parent": arr(d.node_parent, (nn,), np.int32), "abeg": arr(d.node_atom_begin, (nn,), np.int32),
                "aend": arr(d.node_atom_end, (nn,), np.int32), "rel_origin": arr(d.node_rel_origin, (nn, 3), np.float32),
                "rel_axis": arr(d.node_rel_axis, (nn, 3), np.float32), "pairs": arr(d.pairs, (npairs, 2), np.int32),
                "pair_kind": arr(d.pair_kind, (npairs,), np.int32), "n_movable": d.n_movable, "lig_begin": d.lig_begin,
                "lig_end": d.lig_end, "coords0": arr(pxyz, (na, 3), np.float32),
                "conf0": arr(pconf, (7 + nn - 1,), np.float32), "n_tors": nn - 1, "n_lig_tors": tl, "n_flex_tors": tf,
                "n_flex_movable": nfm, "n_inflex": ninf, "num_tors": nt.value}
        return arr(prx, (nrig, 3), np.float32), arr(prs, (nrig,), np.int32), desc
    finally:
        lib().mi_pdbqt_model_close(h)


def sdf_pose_text(name, elements, coords, bonds, energy, rmsd=-1.0, cnnscore=-1.0, cnnaffinity=0.0, cnnvariance=0.0,
                  atom_index=None, props=()):
    """one pose as gnina writes it to an .sdf (mi_sdf_write_pose)"""
    el = b"".join((e.encode() + b"\0\0")[:2] for e in elements)
    xyz = _f32(coords).reshape(-1, 3)
    b = np.ascontiguousarray(bonds, dtype=np.int32).reshape(-1, 3)
    pr = np.ascontiguousarray([(ord(t), a, v) for t, a, v in props], dtype=np.int32).reshape(-1, 3)
    ai = None if atom_index is None else np.ascontiguousarray(atom_index, dtype=np.int32)
    need = C.c_size_t()
    args = (name.encode(), len(elements), el, _ptr(ai), _ptr(xyz), len(b), _ptr(b), len(pr), _ptr(pr), float(energy),
            float(rmsd), float(cnnscore), float(cnnaffinity), float(cnnvariance))
    f = lib().mi_sdf_write_pose
    f.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    if f(*args, None, 0, C.byref(need)) != MI_OK:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    buf = C.create_string_buffer(need.value)
    if f(*args, buf, need.value, C.byref(need)) != MI_OK:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    return buf.value.decode()


def _ligand_num_tors(h):
    nt = C.c_float()
    check(lib().mi_pdbqt_ligand_num_tors(h, C.byref(nt)))
    return nt.value


def pdbqt_poses_text(path_or_text, poses, energies, cnnscores=None, cnnaffinities=None, rmsds=None, is_text=False):
    """Poses [n, n_atoms, 3] (model order) -> gnina's multi-MODEL .pdbqt text (result_info::write)"""
    h = lib().mi_pdbqt_ligand_open(path_or_text.encode(), 1 if is_text else 0)
    if not h:
        raise MiGninaError(lib().mi_pdbqt_last_error().decode())
    try:
        out = []
        for i in range(len(poses)):
            xyz = _f32(poses[i])
            need = C.c_size_t()
            args = (int(i + 1), float(energies[i]), float(-1 if rmsds is None else rmsds[i]),
                    float(-1 if cnnscores is None else cnnscores[i]), float(0 if cnnaffinities is None else cnnaffinities[i]))
            if lib().mi_pdbqt_write_pose(h, _ptr(xyz), *args, None, 0, C.byref(need)) != MI_OK:
                raise MiGninaError(lib().mi_pdbqt_last_error().decode())
            buf = C.create_string_buffer(need.value)
            if lib().mi_pdbqt_write_pose(h, _ptr(xyz), *args, buf, need.value, C.byref(need)) != MI_OK:
                raise MiGninaError(lib().mi_pdbqt_last_error().decode())
            out.append(buf.value.decode())
        return "".join(out)
    finally:
        lib().mi_pdbqt_ligand_close(h)


class Model:
    """One network (mirror of TorchModel's constructor, gninasrc/lib/torch_model.cpp:49-118)."""

    def __init__(self, path_or_name, resolution=None, dimension=None):
        path = path_or_name
        if not os.path.exists(path):
            path = os.path.join(WEIGHTS_DIR, path_or_name.replace(".", "_") + ".mgw")
        if resolution is None and dimension is None:
            self.handle = lib().mi_model_load_file(path.encode())
        else:  # the same weights on another grid (dynamic-pool Dense family only)
            self.handle = lib().mi_model_load_file_ex(path.encode(), float(resolution or 0), float(dimension or 0))
        if not self.handle:
            raise MiGninaError(f"could not load model {path_or_name}: {lib().mi_last_error().decode()}")
        res, dim = C.c_float(), C.c_float()
        nr, nl, n = C.c_int(), C.c_int(), C.c_int()
        check(lib().mi_model_info(self.handle, C.byref(res), C.byref(dim), C.byref(nr), C.byref(nl), C.byref(n)))
        self.resolution, self.dimension = res.value, dim.value
        self.n_rec_channels, self.n_lig_channels, self.grid_points = nr.value, nl.value, n.value
        self.name = lib().mi_model_name(self.handle).decode()

    @property
    def n_channels(self):
        return self.n_rec_channels + self.n_lig_channels

    def type_channel(self, is_ligand, smt):
        r = C.c_float()
        c = lib().mi_model_type_channel(self.handle, int(is_ligand), int(smt), C.byref(r))
        return c, r.value

    def chan_of_smt(self, is_ligand):
        return np.array([self.type_channel(is_ligand, t)[0] for t in range(28)], dtype=np.int32)

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.mi_model_release(self.handle)
            self.handle = None


class Scorer:
    """An ensemble bound to one receptor (mirror of CNNTorchScorer, cnn_torch_scorer.cpp:24-198)."""

    def __init__(self, models):
        self.models = [m if isinstance(m, Model) else Model(m) for m in models]
        arr = (C.c_void_p * len(self.models))(*[m.handle for m in self.models])
        self.handle = lib().mi_scorer_create(arr, len(self.models))
        if not self.handle:
            raise MiGninaError(lib().mi_last_error().decode())

    def set_receptor(self, xyz, smt):
        xyz = _f32(xyz).reshape(-1, 3)
        smt = _i32(smt)
        assert len(xyz) == len(smt)
        check(lib().mi_scorer_set_receptor(self.handle, _ptr(xyz), _ptr(smt), len(smt)))

    def score_batch(self, lig_xyz, lig_smt, centers=None, flags=0):
        """lig_xyz [B][L][3] -> dict(pose, affinity, loss, variance), each float32 [B]."""
        lig_xyz = _f32(lig_xyz)
        B, L = lig_xyz.shape[0], lig_xyz.shape[1]
        lig_smt = _i32(lig_smt)
        assert len(lig_smt) == L
        centers = _f32(centers)
        pose, aff, loss, var = (np.empty(B, dtype=np.float32) for _ in range(4))
        check(lib().mi_scorer_score_batch_ex(self.handle, _ptr(lig_xyz), _ptr(lig_smt), B, L, _ptr(centers),
                                             _ptr(pose), _ptr(aff), _ptr(loss), _ptr(var), flags))
        return {"pose": pose, "affinity": aff, "loss": loss, "variance": var}

    def score_ragged(self, lig_xyz, lig_smt, centers=None):
        """Poses of different ligands in one batch: lig_xyz [B][Lmax][3], lig_smt [B][Lmax] (-1 = padding)."""
        lig_xyz = _f32(lig_xyz)
        B, L = lig_xyz.shape[0], lig_xyz.shape[1]
        lig_smt = _i32(lig_smt)
        assert lig_smt.shape == (B, L)
        centers = _f32(centers)
        pose, aff, loss, var = (np.empty(B, dtype=np.float32) for _ in range(4))
        check(lib().mi_scorer_score_ragged(self.handle, _ptr(lig_xyz), _ptr(lig_smt), B, L, _ptr(centers), _ptr(pose),
                                           _ptr(aff), _ptr(loss), _ptr(var)))
        return {"pose": pose, "affinity": aff, "loss": loss, "variance": var}

    def score_grad(self, lig_xyz, lig_smt, centers=None):
        """Scores + d loss / d ligand coordinates [B][L][3] (refinement path)."""
        lig_xyz = _f32(lig_xyz)
        B, L = lig_xyz.shape[0], lig_xyz.shape[1]
        lig_smt = _i32(lig_smt)
        centers = _f32(centers)
        pose, aff, loss, var = (np.empty(B, dtype=np.float32) for _ in range(4))
        grad = np.empty((B, L, 3), dtype=np.float32)
        check(lib().mi_scorer_score_grad(self.handle, _ptr(lig_xyz), _ptr(lig_smt), B, L, _ptr(centers), _ptr(pose),
                                         _ptr(aff), _ptr(loss), _ptr(var), _ptr(grad)))
        return {"pose": pose, "affinity": aff, "loss": loss, "variance": var, "lig_grad": grad}

    def set_precision(self, bf16):
        """False / 0 / "fp32": the parity path (split-fp16 forward convolutions where planned); True / 1 / "bf16": the bf16
        path; 2 / "fp32_mfma": fp32 MFMA for every layer (include/mi_gnina.h, mi_scorer_set_precision)."""
        code = {"fp32": 0, "bf16": 1, "fp32_mfma": 2, "fp16": 3}.get(bf16, bf16)
        check(lib().mi_scorer_set_precision(self.handle, int(code)))

    def set_rotations(self, quats):
        """per-pose unit quaternions (a, b, c, d) for the NEXT scoring call (TorchModel::forward's `rotate`)"""
        q = _f32(quats).reshape(-1, 4)
        check(lib().mi_scorer_set_rotations(self.handle, _ptr(q), len(q)))

    def set_flex(self, rec_rows):
        """Receptor rows (of set_receptor's arrays) whose coordinates are supplied per pose."""
        rec_rows = _i32(rec_rows)
        self._n_flex = len(rec_rows)
        check(lib().mi_scorer_set_flex(self.handle, _ptr(rec_rows), len(rec_rows)))

    def score_flex(self, lig_xyz, lig_smt, flex_xyz, centers=None, grad=True):
        """Scores with per-pose flexible-residue coordinates [B][n_flex][3]; with grad also
        d loss / d ligand and d loss / d flexible-atom coordinates."""
        lig_xyz = _f32(lig_xyz)
        B, L = lig_xyz.shape[0], lig_xyz.shape[1]
        lig_smt = _i32(lig_smt)
        centers = _f32(centers)
        flex_xyz = _f32(flex_xyz)
        nf = flex_xyz.shape[1]
        pose, aff, loss, var = (np.empty(B, dtype=np.float32) for _ in range(4))
        lg = np.empty((B, L, 3), dtype=np.float32) if grad else None
        fg = np.empty((B, nf, 3), dtype=np.float32) if grad else None
        check(lib().mi_scorer_score_flex(self.handle, _ptr(lig_xyz), _ptr(lig_smt), B, L, _ptr(centers), _ptr(flex_xyz),
                                         _ptr(pose), _ptr(aff), _ptr(loss), _ptr(var), _ptr(lg), _ptr(fg)))
        return {"pose": pose, "affinity": aff, "loss": loss, "variance": var, "lig_grad": lg, "flex_grad": fg}

    def last_model_outputs(self, m, B):
        pose, aff, loss = (np.empty(B, dtype=np.float32) for _ in range(3))
        check(lib().mi_scorer_last_model_outputs(self.handle, m, _ptr(pose), _ptr(aff), _ptr(loss), B))
        return pose, aff, loss

    def voxelize_batch(self, lig_xyz, lig_smt, centers=None, model=0, flags=0):
        """GridMaker::forward for model `model`: returns (grids [B,C,N,N,N], centers [B,3])."""
        lig_xyz = _f32(lig_xyz)
        B, L = lig_xyz.shape[0], lig_xyz.shape[1]
        lig_smt = _i32(lig_smt)
        centers = _f32(centers)
        m = self.models[model]
        N = m.grid_points
        grids = np.empty((B, m.n_channels, N, N, N), dtype=np.float32)
        cen = np.empty((B, 3), dtype=np.float32)
        check(lib().mi_voxelize_batch(self.handle, model, _ptr(lig_xyz), _ptr(lig_smt), B, L, _ptr(centers),
                                      _ptr(grids), _ptr(cen), flags))
        return grids, cen

    def forward_grids(self, grids, model=0):
        grids = _f32(grids)
        B = grids.shape[0]
        pose, aff, loss = (np.empty(B, dtype=np.float32) for _ in range(3))
        check(lib().mi_model_forward_grids(self.handle, model, _ptr(grids), B, _ptr(pose), _ptr(aff), _ptr(loss)))
        return pose, aff, loss

    def score_batch_device(self, lig_ptr, lig_smt, B, L, pose_ptr, aff_ptr, loss_ptr, var_ptr=None, centers_ptr=None):
        """Device-resident inputs/outputs (raw pointers), asynchronous on the scorer's stream."""
        lig_smt = _i32(lig_smt)
        check(lib().mi_scorer_score_batch_ex(self.handle, C.c_void_p(lig_ptr), _ptr(lig_smt), B, L,
                                             C.c_void_p(centers_ptr) if centers_ptr else None,
                                             C.c_void_p(pose_ptr), C.c_void_p(aff_ptr), C.c_void_p(loss_ptr),
                                             C.c_void_p(var_ptr) if var_ptr else None,
                                             MI_LIG_ON_DEVICE | MI_OUT_ON_DEVICE))

    def read_activation(self, buf, B, model_index=0):
        """diagnostic: activation buffer `buf` of the last forward call as fp32 [B][S][S][S][C] + whether it was split
        (mi_debug_read_activation)"""
        info = (C.c_int32 * 3)()
        check(lib().mi_debug_read_activation(self.handle, int(model_index), int(buf), int(B), info, None, 0))
        S, Cc, split = info[0], info[1], info[2]
        out = np.empty((B, S, S, S, Cc), np.float32)
        check(lib().mi_debug_read_activation(self.handle, int(model_index), int(buf), int(B), info,
                                             out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out, bool(split)

    def enable_profile(self, on=True):
        check(lib().mi_scorer_enable_profile(self.handle, int(on)))

    def profile(self):
        import json
        raw = lib().mi_scorer_profile_json(self.handle)
        if raw is None:
            raise MiGninaError(lib().mi_last_error().decode())
        return json.loads(raw.decode())

    def set_chunk(self, n):
        check(lib().mi_scorer_set_chunk(self.handle, int(n)))

    def stream(self):
        return lib().mi_scorer_stream(self.handle)

    def synchronize(self):
        check(lib().mi_scorer_synchronize(self.handle))

    def h2_fallbacks(self):
        """calls repeated on the fp32-MFMA kernels because an activation left the split-fp16 kernels' range"""
        return int(lib().mi_scorer_h2_fallbacks(self.handle))

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.mi_scorer_destroy(self.handle)
            self.handle = None


def user_grid_parse(text):
    """setup_user_gd + the value lines of grid::init (main.cpp:635-670, grid.cpp:69-92) ->
    (begin[3], end[3], n[3], values[nz][ny][nx] float64)"""
    if isinstance(text, str):
        text = text.encode()
    b = np.zeros(3, dtype=np.float32)
    e = np.zeros(3, dtype=np.float32)
    n = np.zeros(3, dtype=np.int32)
    cnt = C.c_size_t(0)
    check(lib().mi_user_grid_parse(text, len(text), _ptr(b), _ptr(e), _ptr(n), None, 0, C.byref(cnt)))
    v = np.zeros(cnt.value, dtype=np.float64)
    check(lib().mi_user_grid_parse(text, len(text), _ptr(b), _ptr(e), _ptr(n), _ptr(v), v.size, C.byref(cnt)))
    return b, e, n, v.reshape(int(n[2]), int(n[1]), int(n[0]))


class Vina:
    """Vina/smina engine: pair tables + receptor cache grids + one prepared ligand
    (mirror of precalculate_linear + cache + model::eval_deriv + quasi_newton)."""

    def __init__(self, weights=None, cutoff=8.0, factor=32.0):
        w = _f32(weights)
        self.handle = lib().mi_vina_create(_ptr(w), cutoff, factor)
        if not self.handle:
            raise MiGninaError(lib().mi_last_error().decode())
        self.n = lib().mi_vina_table_size(self.handle)
        self.grid_shape = None
        self.n_atoms = self.n_tors = 0

    def table(self, t1, t2):
        fast, se, sd = (np.empty(self.n, dtype=np.float32) for _ in range(3))
        check(lib().mi_vina_table(self.handle, t1, t2, _ptr(fast), _ptr(se), _ptr(sd)))
        return fast, se, sd

    def set_receptor(self, xyz, smt):
        xyz, smt = _f32(xyz).reshape(-1, 3), _i32(smt)
        check(lib().mi_vina_set_receptor(self.handle, _ptr(xyz), _ptr(smt), len(smt)))

    def build_cache(self, begin, end, n, lig_types, slope=1e3):
        begin, end, n, lt = _f32(begin), _f32(end), _i32(n), _i32(lig_types)
        check(lib().mi_vina_build_cache(self.handle, _ptr(begin), _ptr(end), _ptr(n), _ptr(lt), len(lt), slope))
        self.grid_shape = (int(n[2]) + 1, int(n[1]) + 1, int(n[0]) + 1)

    def set_approximation(self, kind, factor=10.0):
        """--approximation: 0 linear (precalculate_linear(sf, 32)), 1 spline (precalculate_splines(sf, factor)); before
        build_cache"""
        check(lib().mi_vina_set_approximation(self.handle, int(kind), float(factor)))

    def set_line_search(self, accurate, simple=False):
        """--accurate_line_search for every BFGS of this handle (False = fast_line_search, the default);
        simple: --simple_ascent (minimization_params::Simple) instead of bfgs<>"""
        check(lib().mi_vina_set_line_search(self.handle, 2 if simple else 1 if accurate else 0))

    def set_strict_order(self, on=True):
        """energy sums in the reference's order: trajectories bit-identical to the reference's (mi_gnina.h)"""
        check(lib().mi_vina_set_strict_order(self.handle, 1 if on else 0))

    def pair_eval(self, t1, t2, r2):
        """(E, dE/dr / r) of the current approximation for a type pair at squared distances r2"""
        r2 = _f32(r2).ravel()
        e, d = np.zeros(len(r2), np.float32), np.zeros(len(r2), np.float32)
        check(lib().mi_vina_pair_eval(self.handle, int(t1), int(t2), _ptr(r2), len(r2), _ptr(e), _ptr(d)))
        return e, d

    def set_user_grid(self, begin, end, n, values, scaling_factor=1.0):
        """--user_grid: values = the file's numbers ([nz][ny][nx], float64), or None to remove it; before build_cache"""
        if values is None:
            check(lib().mi_vina_set_user_grid(self.handle, None, None, None, None, 1.0))
            return
        b = np.ascontiguousarray(begin, dtype=np.float32)
        e = np.ascontiguousarray(end, dtype=np.float32)
        nn = np.ascontiguousarray(n, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        assert v.size == int(nn[0]) * int(nn[1]) * int(nn[2])
        check(lib().mi_vina_set_user_grid(self.handle, _ptr(b), _ptr(e), _ptr(nn), _ptr(v), float(scaling_factor)))

    def cache_grid(self, smt):
        out = np.empty(self.grid_shape, dtype=np.float32)
        check(lib().mi_vina_cache_grid(self.handle, int(smt), _ptr(out), out.size))
        return out

    def set_ligand(self, lig):
        a = {k: np.ascontiguousarray(lig[k]) for k in
             ("smt", "local_xyz", "parent", "abeg", "aend", "rel_origin", "rel_axis", "pairs")}
        if lig.get("pair_kind") is not None:
            a["pair_kind"] = np.ascontiguousarray(lig["pair_kind"], dtype=np.int32)
        self._keep = a
        d = LigandDesc(len(a["smt"]), _ptr(a["smt"]), _ptr(a["local_xyz"]), len(a["parent"]), _ptr(a["parent"]),
                       _ptr(a["abeg"]), _ptr(a["aend"]), _ptr(a["rel_origin"]), _ptr(a["rel_axis"]),
                       len(a["pairs"]), _ptr(a["pairs"]), int(lig.get("n_movable", 0)),
                       _ptr(a["pair_kind"]) if "pair_kind" in a else None, int(lig.get("lig_begin", 0)),
                       int(lig.get("lig_end", 0)))
        check(lib().mi_vina_set_ligand(self.handle, C.byref(d)))
        self.n_atoms, self.n_tors = len(a["smt"]), len(a["parent"]) - 1

    def set_screen(self, ligs):
        """mi_vina_set_screen: the ligands of a screen, resident together (see mc_screen)"""
        keep, descs = [], (LigandDesc * len(ligs))()
        for i, lig in enumerate(ligs):
            a = {k: np.ascontiguousarray(lig[k]) for k in
                 ("smt", "local_xyz", "parent", "abeg", "aend", "rel_origin", "rel_axis", "pairs")}
            keep.append(a)
            descs[i] = LigandDesc(len(a["smt"]), _ptr(a["smt"]), _ptr(a["local_xyz"]), len(a["parent"]), _ptr(a["parent"]),
                                  _ptr(a["abeg"]), _ptr(a["aend"]), _ptr(a["rel_origin"]), _ptr(a["rel_axis"]),
                                  len(a["pairs"]), _ptr(a["pairs"]))
        check(lib().mi_vina_set_screen(self.handle, len(ligs), C.cast(descs, C.c_void_p)))
        mc, mh, ma = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().mi_vina_screen_dims(self.handle, C.byref(mc), C.byref(mh), C.byref(ma)))
        self.screen_conf, self.screen_heavy, self.screen_atoms = mc.value, mh.value, ma.value
        self.screen_natoms = [len(a["smt"]) for a in keep]
        self.screen_tors = [len(a["parent"]) - 1 for a in keep]
        self.screen_nheavy = [int((a["smt"] > 1).sum()) for a in keep]

    def mc_screen(self, chain_ligand, seeds, corner1, corner2, params):
        """mi_vina_mc_screen: chain b docks ligand chain_ligand[b]; params = one McParams per ligand.
        -> (n [B], e [B,S], confs [B,S,max_conf], coords [B,S,max_heavy,3], evals [B]); a chain of ligand l fills
        confs[..., :7+T_l] and coords[..., :3*n_heavy_l] of the flattened last two axes."""
        chain_ligand = np.ascontiguousarray(chain_ligand, dtype=np.int32)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        B, S = len(seeds), params[0].num_saved
        P = (McParams * len(params))(*params)
        c1, c2 = _f32(corner1), _f32(corner2)
        n = np.zeros(B, dtype=np.int32)
        e = np.zeros((B, S), dtype=np.float32)
        cf = np.zeros((B, S, self.screen_conf), dtype=np.float32)
        xyz = np.zeros((B, S, self.screen_heavy * 3), dtype=np.float32)
        ev = np.zeros(B, dtype=np.int32)
        check(lib().mi_vina_mc_screen(self.handle, B, _ptr(chain_ligand), _ptr(seeds), _ptr(c1), _ptr(c2),
                                      C.cast(P, C.c_void_p), _ptr(n), _ptr(e), _ptr(cf), _ptr(xyz), _ptr(ev)))
        return n, e, cf, xyz, ev

    def eval_screen(self, item_ligand, confs, v=(1000.0, 1000.0, 1000.0), deriv=True, want_coords=False):
        """mi_vina_eval_screen: confs [B, max_conf] -> (e [B], change [B, max_conf-1] or None, coords [B, max_atoms, 3] or None)"""
        item_ligand = np.ascontiguousarray(item_ligand, dtype=np.int32)
        confs = _f32(confs).reshape(-1, self.screen_conf)
        B = len(confs)
        vv = _f32(v)
        e = np.empty(B, dtype=np.float32)
        ch = np.zeros((B, self.screen_conf - 1), dtype=np.float32) if deriv else None
        co = np.zeros((B, self.screen_atoms, 3), dtype=np.float32) if want_coords else None
        check(lib().mi_vina_eval_screen(self.handle, _ptr(item_ligand), _ptr(confs), B, _ptr(vv), int(deriv), _ptr(e),
                                        _ptr(ch), _ptr(co)))
        return e, ch, co

    def refine_screen(self, item_ligand, confs, max_iters, v=(1000.0, 1000.0, 1000.0)):
        """mi_vina_refine_screen: confs [B, max_conf] (copied), max_iters per ligand -> (e, confs, tries)"""
        item_ligand = np.ascontiguousarray(item_ligand, dtype=np.int32)
        confs = np.array(_f32(confs).reshape(-1, self.screen_conf), copy=True)
        B = len(confs)
        mi = np.ascontiguousarray(max_iters, dtype=np.int32)
        vv = _f32(v)
        e = np.empty(B, dtype=np.float32)
        tries = np.empty(B, dtype=np.int32)
        check(lib().mi_vina_refine_screen(self.handle, _ptr(item_ligand), _ptr(confs), B, _ptr(vv), _ptr(mi), _ptr(e),
                                          _ptr(tries)))
        return e, confs, tries

    def final_energies_screen(self, item_ligand, confs, num_tors, v=(1000.0, 1000.0, 1000.0)):
        item_ligand = np.ascontiguousarray(item_ligand, dtype=np.int32)
        confs = _f32(confs).reshape(-1, self.screen_conf)
        B = len(confs)
        nt = _f32(num_tors)
        vv = _f32(v)
        e, intra = np.empty(B, dtype=np.float32), np.empty(B, dtype=np.float32)
        check(lib().mi_vina_final_energies_screen(self.handle, _ptr(item_ligand), _ptr(confs), B, _ptr(vv), _ptr(nt),
                                                  _ptr(e), _ptr(intra)))
        return e, intra

    def eval_latency_us(self, confs, mode, reps=200):
        confs = _f32(confs).reshape(-1, 7 + self.n_tors)
        ms = C.c_float()
        check(lib().mi_vina_eval_latency(self.handle, _ptr(confs), len(confs), mode, reps, C.byref(ms)))
        return 1e3 * ms.value / reps

    def refine_batch(self, confs, v=(1000.0, 1000.0, 1000.0), max_iters=None):
        """refine_structure: BFGS on the direct receptor term with the slope ladder -> (e, confs, tries)"""
        confs = np.array(_f32(confs).reshape(-1, 7 + self.n_tors), copy=True)
        B = len(confs)
        if max_iters is None:
            max_iters = (25 + self.n_atoms) // 3
        vv = _f32(v)
        e = np.empty(B, dtype=np.float32)
        tries = np.empty(B, dtype=np.int32)
        check(lib().mi_vina_refine_batch(self.handle, _ptr(confs), B, _ptr(vv), int(max_iters), _ptr(e), _ptr(tries)))
        return e, confs, tries

    def cache_eval_coords(self, coords, smt, v=1000.0, deriv=True):
        """igrid::eval / eval_deriv of the cache on coordinates [B][n][3] -> (energy [B], minus_forces [B][n][3] | None)"""
        coords = _f32(coords)
        if coords.ndim == 2:
            coords = coords[None]
        B, n = coords.shape[0], coords.shape[1]
        smt = _i32(smt)
        e = np.empty(B, dtype=np.float32)
        f = np.empty((B, n, 3), dtype=np.float32) if deriv else None
        check(lib().mi_vina_cache_eval_coords(self.handle, _ptr(coords), _ptr(smt), n, B, float(v), _ptr(e), _ptr(f)))
        return e, f

    def coords_batch(self, confs):
        """model::set(conf): coords [B, n_atoms, 3]"""
        confs = _f32(confs).reshape(-1, 7 + self.n_tors)
        out = np.empty((len(confs), self.n_atoms, 3), dtype=np.float32)
        check(lib().mi_vina_coords_batch(self.handle, _ptr(confs), len(confs), _ptr(out)))
        return out

    def cnn_eval_batch(self, scorer, confs, box, cnn_centers=None, deriv=True):
        """non_cache_cnn::eval_deriv / eval -> (energy [B], change [B, 6+T] or None)"""
        confs = _f32(confs).reshape(-1, 7 + self.n_tors)
        B = len(confs)
        cen = _f32(cnn_centers)
        e = np.empty(B, dtype=np.float32)
        ch = np.empty((B, 6 + self.n_tors), dtype=np.float32) if deriv else None
        check(lib().mi_cnn_eval_batch(self.handle, scorer.handle, _ptr(confs), B, C.byref(box), _ptr(cen),
                                      1 if deriv else 0, _ptr(e), _ptr(ch)))
        return e, ch

    def cnn_refine_batch(self, scorer, confs, box, max_iters=None):
        """refine_structure with non_cache_cnn -> (energy, confs, tries, evals)"""
        confs = np.array(_f32(confs).reshape(-1, 7 + self.n_tors), copy=True)
        B = len(confs)
        if max_iters is None:
            max_iters = (25 + self.n_atoms) // 3
        e = np.empty(B, dtype=np.float32)
        tries, evals = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        check(lib().mi_cnn_refine_batch(self.handle, scorer.handle, _ptr(confs), B, C.byref(box), int(max_iters),
                                        _ptr(e), _ptr(tries), _ptr(evals)))
        return e, confs, tries, evals

    def final_energies(self, confs, num_tors, v=(1000.0, 1000.0, 1000.0)):
        confs = _f32(confs).reshape(-1, 7 + self.n_tors)
        B = len(confs)
        vv = _f32(v)
        e, intra = np.empty(B, dtype=np.float32), np.empty(B, dtype=np.float32)
        check(lib().mi_vina_final_energies(self.handle, _ptr(confs), B, _ptr(vv), float(num_tors), _ptr(e), _ptr(intra)))
        return e, intra

    def eval_batch(self, confs, v=(1000.0, 1000.0, 1000.0), deriv=True, want_coords=False, grid_only=False,
                   direct=False, exact=False, pairs_only=False):
        confs = _f32(confs).reshape(-1, 7 + self.n_tors)
        B = len(confs)
        vv = _f32(v)
        e = np.empty(B, dtype=np.float32)
        ch = np.empty((B, 6 + self.n_tors), dtype=np.float32) if (deriv and not grid_only and not pairs_only) else None
        co = np.empty((B, self.n_atoms, 3), dtype=np.float32) if want_coords else None
        mode = 4 if pairs_only else (2 if grid_only else int(deriv))   # 1 eval_deriv, 0 eval, 2 receptor only, 4 pairs only
        mode |= (16 if direct else 0) | (32 if exact else 0)
        check(lib().mi_vina_eval_batch(self.handle, _ptr(confs), B, _ptr(vv), mode, _ptr(e), _ptr(ch), _ptr(co)))
        return e, ch, co

    def mc_batch(self, seeds, corner1, corner2, params):
        """B Monte-Carlo chains -> (n_saved [B], energies [B,S], confs [B,S,7+T], coords [B,S,nh,3], evals [B])"""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        B, S = len(seeds), params.num_saved
        nh = lib().mi_vina_ligand_heavy_atoms(self.handle)
        c1, c2 = _f32(corner1), _f32(corner2)
        n = np.zeros(B, dtype=np.int32)
        e = np.zeros((B, S), dtype=np.float32)
        cf = np.zeros((B, S, 7 + self.n_tors), dtype=np.float32)
        xyz = np.zeros((B, S, nh, 3), dtype=np.float32)
        ev = np.zeros(B, dtype=np.int32)
        check(lib().mi_vina_mc_batch(self.handle, B, _ptr(seeds), _ptr(c1), _ptr(c2), C.byref(params), _ptr(n),
                                     _ptr(e), _ptr(cf), _ptr(xyz), _ptr(ev)))
        return n, e, cf, xyz, ev

    def mc_cnn_batch(self, scorer, seeds, corner1, corner2, params, box, level_all=False):
        """Monte-Carlo chains with the CNN as the Metropolis energy (metrorescore / metrorefine), or -- level_all --
        as the igrid of the minimiser too (--cnn_scoring all) ->
        (n_saved [B], energies [B,S], confs [B,S,7+T], coords [B,S,nh,3], evals [B], cnn_evals)"""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        B, S = len(seeds), params.num_saved
        nh = lib().mi_vina_ligand_heavy_atoms(self.handle)
        c1, c2 = _f32(corner1), _f32(corner2)
        n = np.zeros(B, dtype=np.int32)
        e = np.zeros((B, S), dtype=np.float32)
        cf = np.zeros((B, S, 7 + self.n_tors), dtype=np.float32)
        xyz = np.zeros((B, S, nh, 3), dtype=np.float32)
        ev = np.zeros(B, dtype=np.int32)
        ce = C.c_int32()
        fn = lib().mi_vina_mc_cnnall_batch if level_all else lib().mi_vina_mc_cnn_batch
        check(fn(self.handle, scorer.handle, B, _ptr(seeds), _ptr(c1), _ptr(c2), C.byref(params),
                 C.byref(box), _ptr(n), _ptr(e), _ptr(cf), _ptr(xyz), _ptr(ev), C.byref(ce)))
        return n, e, cf, xyz, ev, ce.value

    def bfgs_batch(self, confs, v=(1000.0, 1000.0, 1000.0), max_iters=None):
        confs = np.array(_f32(confs).reshape(-1, 7 + self.n_tors), copy=True)
        B = len(confs)
        if max_iters is None:
            max_iters = (25 + self.n_atoms) // 3
        vv = _f32(v)
        e = np.empty(B, dtype=np.float32)
        g = np.empty((B, 6 + self.n_tors), dtype=np.float32)
        ev = np.empty(B, dtype=np.int32)
        check(lib().mi_vina_bfgs_batch(self.handle, _ptr(confs), B, _ptr(vv), int(max_iters), _ptr(e), _ptr(g),
                                       _ptr(ev)))
        return e, confs, g, ev

    @classmethod
    def _borrow(cls, handle):
        """a view on a handle someone else owns (VinaPool's per-device handles)"""
        v = cls.__new__(cls)
        v.handle = handle
        v._borrowed = True
        v.n = lib().mi_vina_table_size(handle)
        v.grid_shape = None
        v.n_atoms = v.n_tors = 0
        return v

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None and not getattr(self, "_borrowed", False):
            _lib.mi_vina_destroy(self.handle)
            self.handle = None


def merge_mc_outputs(n, e, conf, coords, min_rmsd=2.0, max_size=50):
    """parallel_mc's merge of the per-chain containers -> (energies [m], confs [m,nc], coords [m,nh,3])"""
    n = np.ascontiguousarray(n, dtype=np.int32)
    e, conf, coords = _f32(e), _f32(conf), _f32(coords)
    B, S = e.shape
    nc, nh = conf.shape[2], coords.shape[2]
    on = C.c_int32()
    oe = np.empty(max_size, dtype=np.float32)
    ocf = np.empty((max_size, nc), dtype=np.float32)
    oxyz = np.empty((max_size, nh, 3), dtype=np.float32)
    check(lib().mi_merge_mc_outputs(_ptr(n), _ptr(e), _ptr(conf), _ptr(coords), B, S, nc, nh, min_rmsd, max_size,
                                    C.byref(on), _ptr(oe), _ptr(ocf), _ptr(oxyz)))
    m = on.value
    return oe[:m].copy(), ocf[:m].copy(), oxyz[:m].copy()


def rank_poses(cnnscore, cnnaffinity, energy, coords, sort_order=0, min_rmsd=1.0):
    """do_search's sort + remove_redundant (host only): returns the kept pose indices, best first."""
    coords = _f32(coords)
    n, nh = coords.shape[0], coords.shape[1]
    cs, ca, en = _f32(cnnscore), _f32(cnnaffinity), _f32(energy)
    order = np.empty(max(n, 1), dtype=np.int32)
    n_out = C.c_int32()
    check(lib().mi_rank_poses(_ptr(cs), _ptr(ca), _ptr(en), _ptr(coords), n, nh, sort_order, min_rmsd, _ptr(order),
                              C.byref(n_out)))
    return order[:n_out.value].copy()


class Pool:
    """mi_pool: one scorer per listed GPU in THIS process (C++ worker threads), pose shards run concurrently.
    models: names of built-in blobs or paths."""

    def __init__(self, models, devices=None):
        if devices is None:
            devices = list(range(lib().mi_gnina_device_count()))
        paths = [m if os.path.exists(m) else os.path.join(_HERE, "weights", m + ".mgw") for m in models]
        dv = (C.c_int * len(devices))(*devices)
        pv = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        self.handle = lib().mi_pool_create(dv, len(devices), pv, len(paths))
        if not self.handle:
            raise MiGninaError(lib().mi_last_error().decode())
        self.devices = list(devices)

    def set_receptor(self, xyz, smt):
        xyz, smt = _f32(xyz), _i32(smt)
        check(lib().mi_pool_set_receptor(self.handle, _ptr(xyz), _ptr(smt), len(smt)))

    def score_batch(self, lig_xyz, lig_smt, centers=None):
        lig_xyz, lig_smt, centers = _f32(lig_xyz), _i32(lig_smt), _f32(centers)
        B, L = lig_xyz.shape[0], lig_xyz.shape[1]
        o = [np.empty(B, dtype=np.float32) for _ in range(4)]
        check(lib().mi_pool_score_batch(self.handle, _ptr(lig_xyz), _ptr(lig_smt), B, L, _ptr(centers), _ptr(o[0]),
                                        _ptr(o[1]), _ptr(o[2]), _ptr(o[3]), 0))
        return {"pose": o[0], "affinity": o[1], "loss": o[2], "variance": o[3]}

    def score_batch_device(self, lig_ptr, lig_smt, B, L, pose_ptr, aff_ptr, loss_ptr, var_ptr=None, centers_ptr=None):
        """poses / outputs resident on devices[0] (raw pointers): RCCL scatter / gather when the pool has > 1 device"""
        lig_smt = _i32(lig_smt)
        check(lib().mi_pool_score_batch(self.handle, lig_ptr, _ptr(lig_smt), B, L, centers_ptr, pose_ptr, aff_ptr, loss_ptr,
                                        var_ptr, 3))

    def score_ragged(self, lig_xyz, lig_smt, centers=None):
        lig_xyz, lig_smt, centers = _f32(lig_xyz), _i32(lig_smt), _f32(centers)
        B, L = lig_xyz.shape[0], lig_xyz.shape[1]
        o = [np.empty(B, dtype=np.float32) for _ in range(4)]
        check(lib().mi_pool_score_ragged(self.handle, _ptr(lig_xyz), _ptr(lig_smt), B, L, _ptr(centers), _ptr(o[0]),
                                         _ptr(o[1]), _ptr(o[2]), _ptr(o[3])))
        return {"pose": o[0], "affinity": o[1], "loss": o[2], "variance": o[3]}

    def info(self):
        import json
        return json.loads(lib().mi_pool_info_json(self.handle).decode())

    def __del__(self):
        if getattr(self, "handle", None):
            lib().mi_pool_destroy(self.handle)
            self.handle = None


class VinaPool:
    """mi_vina_pool: one mi_vina handle per listed GPU in THIS process; Monte-Carlo chains split by chain id (a screen's
    by ligand), set-up replicated through configure()."""

    _CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)

    def __init__(self, devices=None, weights=None, cutoff=8.0, factor=32.0):
        L = lib()
        L.mi_vina_pool_create.restype = C.c_void_p
        L.mi_vina_pool_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float]
        L.mi_vina_pool_destroy.argtypes = [C.c_void_p]
        L.mi_vina_pool_destroy.restype = None
        L.mi_vina_pool_size.argtypes = [C.c_void_p]
        L.mi_vina_pool_configure.argtypes = [C.c_void_p, self._CB, C.c_void_p]
        L.mi_vina_pool_mc_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 5
        L.mi_vina_pool_mc_screen.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int] + [C.c_void_p] * 5
        L.mi_vina_pool_info_json.argtypes = [C.c_void_p]
        L.mi_vina_pool_info_json.restype = C.c_char_p
        if devices is None:
            devices = list(range(L.mi_gnina_device_count()))
        dv = (C.c_int * len(devices))(*devices)
        w = _f32(weights)
        self.handle = L.mi_vina_pool_create(dv, len(devices), _ptr(w), cutoff, factor)
        if not self.handle:
            raise MiGninaError(L.mi_last_error().decode())
        self.devices = list(devices)
        self.views = {}

    def configure(self, fn):
        """fn(vina, rank) once per device, on that device's worker thread: the same set-up calls for every rank"""
        errors = []

        def tramp(h, rank, _user):
            try:
                if rank not in self.views:
                    self.views[rank] = Vina._borrow(h)
                fn(self.views[rank], rank)
                return 0
            except Exception as e:  # noqa: BLE001 (reported to the caller below)
                errors.append(e)
                return 1

        cb = self._CB(tramp)
        st = lib().mi_vina_pool_configure(self.handle, cb, None)
        if errors:
            raise errors[0]
        check(st)

    def mc_batch(self, seeds, corner1, corner2, params):
        """Vina.mc_batch over the pool: chains split by chain id"""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        v0 = self.views[0]
        B, S = len(seeds), params.num_saved
        nh = lib().mi_vina_ligand_heavy_atoms(v0.handle)
        cs = 7 + v0.n_tors
        c1, c2 = _f32(corner1), _f32(corner2)
        n, ev = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        e = np.zeros((B, S), dtype=np.float32)
        cf = np.zeros((B, S, cs), dtype=np.float32)
        xyz = np.zeros((B, S, nh, 3), dtype=np.float32)
        check(lib().mi_vina_pool_mc_batch(self.handle, B, _ptr(seeds), _ptr(c1), _ptr(c2), C.addressof(params), cs, nh,
                                          _ptr(n), _ptr(e), _ptr(cf), _ptr(xyz), _ptr(ev)))
        return n, e, cf, xyz, ev

    def info(self):
        import json
        return json.loads(lib().mi_vina_pool_info_json(self.handle).decode())

    def __del__(self):
        if getattr(self, "handle", None):
            self.views = {}
            lib().mi_vina_pool_destroy(self.handle)
            self.handle = None


def device_libm(x):
    """(sinf, cosf, expf, logf(|x|)) of float32 x as the Vina kernels compute them: glibc's algorithms restated in fp64
    (vina.hip sincos_ref / expf_ref / logf_ref) so that they give the bits of the reference's host libm"""
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    f32p = C.POINTER(C.c_float)
    o = [np.empty_like(x) for _ in range(4)]
    p = [a.ctypes.data_as(f32p) for a in [x] + o]
    check(lib().mi_debug_sincos(p[0], len(x), p[1], p[2]))
    check(lib().mi_debug_explog(p[0], len(x), p[3], p[4]))
    return tuple(o)


def split_f16(x, scale=0.0):
    """(hi, lo, scale): the fp16 operand split of the split-fp16 conv kernels as the model loader applies it to weights
    (host code, no device needed); hi / lo are float16 arrays."""
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    hi, lo = np.empty(len(x), dtype=np.uint16), np.empty(len(x), dtype=np.uint16)
    used = C.c_float()
    u16p, f32p = C.POINTER(C.c_uint16), C.POINTER(C.c_float)
    check(lib().mi_debug_split_f16(x.ctypes.data_as(f32p), len(x), float(scale), hi.ctypes.data_as(u16p),
                                   lo.ctypes.data_as(u16p), C.byref(used)))
    return hi.view(np.float16), lo.view(np.float16), used.value


def device_acosf(x):
    """acosf of float32 x as the accurate line search's quaternion_to_angle computes it on the device"""
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    f32p = C.POINTER(C.c_float)
    o = np.empty_like(x)
    check(lib().mi_debug_acos(x.ctypes.data_as(f32p), len(x), o.ctypes.data_as(f32p)))
    return o


def init(device=0):
    check(lib().mi_gnina_init(device))


def set_option(name, value=None):
    """mi_gnina_set_option: an experiment / A-B switch of the library (gnina_amd/csrc/options.h).  The library reads the
    environment once per process; afterwards this is the only way to change a switch.  value None = unset."""
    check(lib().mi_gnina_set_option(name.encode(), None if value is None else str(value).encode()))


class option:
    """with capi.option("MI_GNINA_H2_WLDS", 0): ... -- a switch set for the block, restored to unset afterwards"""

    def __init__(self, name, value="1"):
        self.name, self.value = name, value

    def __enter__(self):
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, None)
        return False


def options():
    return lib().mi_gnina_options().decode()
