"""ctypes binding of libcomo_hip.so (the C ABI declared in include/como_hip.h).

There is NO fallback: if the library is missing or a call fails, the caller gets a RuntimeError.
torch is imported first so that the HIP runtime torch bundles (libamdhip64.so, soname .so.7) is the one
the library binds to -- one runtime, one set of streams.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# COMO_HIP_LIB: measurement scripts point this at the -DCOMO_AB_VARIANTS build (como_amd/lib_ab/libcomo_hip_ab.so); the product path
# is the in-tree library
LIB_PATH = os.environ.get("COMO_HIP_LIB") or os.path.join(_HERE, "lib", "libcomo_hip.so")
_lib = None

c_void_p, c_int, c_long, c_double, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_double, ctypes.c_float


class BAArgs(ctypes.Structure):
    """Mirror of `struct como_ba_args` (include/como_hip.h) -- field order must match."""
    _fields_ = [
        ("b", c_int), ("n", c_int), ("m", c_int), ("H", c_int), ("W", c_int), ("zmode", c_int), ("chunks", c_int),
        ("phase", c_int), ("h_is_f64", c_int), ("variant", c_int), ("stagger", c_int), ("pix_begin", c_int), ("pix_end", c_int),
        ("anorm_f32", c_int),
        ("Pwn", c_void_p), ("vals", c_void_p), ("dPwn_dTwc", c_void_p), ("zjac", c_void_p), ("uvec", c_void_p),
        ("pixidx", c_void_p), ("invz", c_void_p), ("kt_slot_stride", c_long), ("poses_all", c_void_p),
        ("aff_all", c_void_p), ("img_base", c_void_p), ("K", c_void_p), ("ref_slot", c_void_p), ("ref_aff", c_void_p),
        ("tgt_aff", c_void_p), ("tgt_pose", c_void_p), ("tgt_img", c_void_p), ("pose_ref_inds", c_void_p),
        ("pose_tgt_inds", c_void_p), ("landmark_inds", c_void_p), ("dzdP", c_void_p), ("Hmat", c_void_p),
        ("gvec", c_void_p), ("D", c_long), ("err_out", c_void_p), ("sigma_out", c_void_p), ("pj_out", c_void_p),
        ("pair_blocks_out", c_void_p), ("ws_r", c_void_p), ("ws_valid", c_void_p), ("ws_hists", c_void_p),
        ("ws_pair", c_void_p), ("ws_partials", c_void_p), ("grp_pairs", c_void_p), ("single_pairs", c_void_p),
        ("ngrp", c_int), ("nsingle", c_int),
        ("fix_plane", c_long), ("reduce_mode", c_int), ("blocks_fix", c_void_p),
        ("channels", c_int), ("pair_chan", c_void_p), ("ref_pose", c_void_p),
        ("asm_grp_start", c_void_p), ("asm_grp_list", c_void_p), ("n_asm_grp", c_int),
    ]


class DRFuse(ctypes.Structure):
    """Mirror of `struct como_dr_fuse` (include/como_hip.h)."""
    _fields_ = [("ref_pairs", c_void_p), ("np_max", c_int), ("pair_T", c_void_p), ("pair_aff", c_void_p), ("vals", c_void_p),
                ("img_base", c_void_p), ("tgt_img", c_void_p), ("r_out", c_void_p), ("valid_out", c_void_p), ("rhists", c_void_p),
                ("H", c_int), ("W", c_int), ("anorm_f32", c_int)]


class WinArgs(ctypes.Structure):
    """Mirror of `struct como_win_args` (include/como_hip.h) -- field order must match."""
    _fields_ = ([(n, c_int) for n in ("B", "F", "m", "L", "nfix", "pix_is_f64", "median_new_is_f32", "median_new_stride")]
                + [("D", c_long)]
                + [(n, c_void_p) for n in ("poses", "aff", "K", "median", "pm_first", "Kmm_inv", "pose_anchor", "aff_anchor",
                                          "P_anchor", "P_m", "lm_ids", "first_frame", "first_slot", "fix_lm", "first_mask",
                                          "pose_inds", "landmark_inds", "fix_inds", "median_new", "pm", "logzm", "invz", "dzdP",
                                          "dlogz_dT", "dlogz_dP", "dp_dP", "dp_dT", "init_Pm", "reinit_flag", "px_logzm",
                                          "px_invz", "px_dzdP", "px_dlogz_dT", "px_poses", "px_aff")]
                + [(n, c_double) for n in ("s_gp", "s_ld", "s_px", "s_pose", "s_aff", "s_lm")]
                + [(n, c_void_p) for n in ("H", "g", "err")]
                + [("zero_a", c_void_p), ("zero_a_bytes", c_long), ("zero_b", c_void_p), ("zero_b_bytes", c_long),
                   ("median_out", c_void_p), ("sysfix", c_void_p), ("fix_plane", c_long),
                   ("mld_J", c_void_p), ("mld_anchor", c_void_p), ("s_mld", c_double),
                   ("zero_c", c_void_p), ("zero_c_bytes", c_long)])


# name -> (restype, argtypes); every symbol include/como_hip.h declares
SIGNATURES = {
    "como_abi_version": (c_int, []),
    "como_clear_last_error": (c_int, []),
    "como_abort_capture": (c_int, [c_void_p]),
    "como_select_workspace_bytes": (c_int, []),
    "como_select_begin": (c_int, [c_void_p, c_int, c_void_p]),
    "como_select_hist_f32": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_int, c_void_p]),
    "como_select_hist_f64": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_int, c_void_p]),
    "como_select_finish_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "como_select_finish_f64": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "como_select_cand_words": (c_int, []),
    "como_select_cand_pack": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "como_select_cand_merge": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "como_track_partials_bytes": (c_long, []),
    "como_track_iter_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long] + [c_void_p] * 9),
    "como_track_iter_f64": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long] + [c_void_p] * 9),
    "como_track_iter_masked_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long] + [c_void_p] * 10),
    "como_track_iter_masked_f64": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long] + [c_void_p] * 10),
    "como_track_level_workspace_bytes": (c_long, []),
    "como_track_level_workspace_create": (c_void_p, []),
    "como_track_level_workspace_destroy": (None, [c_void_p]),
    "como_track_level_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long, c_void_p, c_void_p, c_int, c_float, c_float, c_float,
                                     c_void_p, c_int, c_void_p, c_void_p]),
    "como_track_iter_channels_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long, c_int] + [c_void_p] * 10),
    "como_track_iter_channels_f64": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long, c_int] + [c_void_p] * 10),
    "como_track_level_channels_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long, c_int, c_void_p, c_void_p, c_int, c_float,
                                              c_float, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "como_track_level_local_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long, c_int, c_void_p, c_void_p, c_int, c_float,
                                           c_float, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "como_track_level_prezeroed_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long, c_int, c_void_p, c_void_p, c_int, c_float,
                                               c_float, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "como_track_level_zero_bytes": (c_long, []),
    "como_track_frame_pyramid3_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "como_track_level_probe": (c_int, []),
    "como_track_level_set_local": (c_int, [c_int]),
    "como_track_level_set_one": (c_int, [c_int]),
    "como_track_level_set_split": (c_int, [c_int]),
    "como_track_level_debug_amb_cap": (None, [c_int]),
    "como_track_level_local_state": (c_int, []),
    "como_track_level_debug_mismatch": (None, [c_int]),
    "como_track_reference_f32": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_float, c_float] + [c_void_p] * 4),
    "como_track_reference_f64": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_double, c_double] + [c_void_p] * 4),
    "como_track_reference_pyr_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 7 + [c_float, c_float, c_void_p]),
    "como_reproject_depth_f32": (c_int, [c_void_p] * 3 + [c_long, c_int, c_int] + [c_void_p] * 6),
    "como_reproject_depth_f64": (c_int, [c_void_p] * 3 + [c_long, c_int, c_int] + [c_void_p] * 6),
    "como_reproject_points_f32": (c_int, [c_void_p] * 4 + [c_long, c_int, c_int, c_int, c_float] + [c_void_p] * 4),
    "como_reproject_points_f64": (c_int, [c_void_p] * 4 + [c_long, c_int, c_int, c_int, c_double] + [c_void_p] * 4),
    "como_ba_partials_elems": (c_long, [c_int, c_int, c_int]),
    "como_sys_fix_plane_elems": (c_long, [c_long]),
    "como_sys_finalize": (c_int, [c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "como_ba_linearize_f32": (c_int, [ctypes.POINTER(BAArgs), c_void_p]),
    "como_ba_linearize_f64": (c_int, [ctypes.POINTER(BAArgs), c_void_p]),
    "como_cross_covariance_f32": (c_int, [c_void_p] * 4 + [c_float, c_void_p, c_int, c_int, c_int,
                                                        ctypes.POINTER(c_long), c_void_p]),
    "como_cross_covariance_f16": (c_int, [c_void_p] * 4 + [c_float, c_void_p, c_int, c_int, c_int,
                                                        ctypes.POINTER(c_long), c_void_p]),
    "como_cross_covariance_f64": (c_int, [c_void_p] * 4 + [c_double, c_void_p, c_int, c_int, c_int,
                                                        ctypes.POINTER(c_long), c_void_p]),
    "como_chol_append_obs_info_f32": (c_int, [c_void_p] * 5 + [c_float, c_int, c_int, c_int, c_int, c_void_p]),
    "como_greedy_loop_f32": (c_int, [c_void_p] * 11 + [c_float, c_float, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "como_greedy_thin_f32": (c_int, [c_void_p] * 11 + [c_float] * 5 + [c_int, c_int, c_void_p, c_void_p]),
    "como_greedy_loop_ws_f32": (c_int, [c_void_p] * 11 + [c_float, c_float, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_long,
                                        c_void_p]),
    "como_greedy_persist_workspace_bytes": (c_long, [c_int, c_int]),
    "como_greedy_persist_f32": (c_int, [c_void_p] * 11 + [c_float, c_float, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "como_greedy_next_f32": (c_int, [c_void_p] * 3 + [c_int, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "como_dense_ref_f32": (c_int, [c_void_p, c_long] + [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 8 + [c_int, c_void_p]),
    "como_dense_ref_f64": (c_int, [c_void_p, c_long] + [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 8 + [c_int, c_void_p]),
    "como_dense_ref_fused_f32": (c_int, [c_void_p, c_long] + [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 8 + [c_int, ctypes.POINTER(DRFuse), c_void_p]),
    "como_dense_ref_fused_f64": (c_int, [c_void_p, c_long] + [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 8 + [c_int, ctypes.POINTER(DRFuse), c_void_p]),
    "como_depth_band_f32": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 8 + [c_int, c_void_p]),
    "como_depth_band_f64": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 8 + [c_int, c_void_p]),
    "como_cov_params_at": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "como_diag_cov_f32": (c_int, [c_void_p, c_long, c_float, c_void_p, c_void_p]),
    "como_diag_cov_f64": (c_int, [c_void_p, c_long, c_double, c_void_p, c_void_p]),
    "como_kernel_matrices_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_float] + [c_void_p] * 7 + [c_int, c_void_p]),
    "como_kernel_matrices_f64": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_double] + [c_void_p] * 7 + [c_int, c_void_p]),
    "como_backproject_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "como_backproject_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "como_kernel_matrix_f32": (c_int, [c_void_p] * 4 + [c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "como_kernel_matrix_f64": (c_int, [c_void_p] * 4 + [c_double, c_void_p, c_int, c_int, c_int, c_void_p]),
    "como_ktilde_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p]),
    "como_ktilde_f64": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p]),
    "como_ktilde_mirror_f64": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    "como_chol_workspace_bytes": (c_long, [c_int]),
    "como_chol_solve_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "como_chol_solve_packed_f64": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "como_chol_set_persistent": (c_int, [c_int]),
    "como_chol_persistent_state": (c_int, []),
    "como_chol_debug_stall": (None, [c_int]),
    "como_sys_finalize_pack": (c_int, [c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "como_chol_small_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "como_chol_small_f64": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "como_trsm_lower_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_int, c_void_p]),
    "como_trsm_lower_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_int, c_void_p]),
    "como_nn_conv2d_f32": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p, c_int, c_void_p]),
    "como_nn_conv2d_fused_f32": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p, c_int] + [c_void_p] * 3 + [c_float, c_void_p]),
    "como_kf_predictor_sinv_f64": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "como_kf_distill_prep_f64": (c_int, [c_void_p, c_long, c_void_p, c_long, c_double, c_void_p, c_double, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "como_kf_corr_good_f64": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_long, c_double, c_double, c_void_p, c_void_p]),
    "como_kf_normalize_coords_f32": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "como_kf_normalize_coords_f64": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "como_kf_normalize_coords_swap_f32": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "como_kf_normalize_coords_swap_f64": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "como_kf_grad_mag_f32": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "como_kf_grad_mag_f64": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "como_kf_masked_std_f64": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "como_kf_cond_c_f64": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "como_kf_cond_system_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "como_kf_aff_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "como_kf_aff_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "como_nn_conv2d_gn_f32": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p,
                                      c_void_p]),
    "como_nn_gn_finalize_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_float, c_void_p, c_void_p]),
    "como_nn_deep_part_floats": (c_long, [c_int] * 5),
    "como_nn_conv3x3_deep_f32": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p, c_float, c_void_p, c_long, c_int, c_void_p, c_void_p,
                                         c_float, c_void_p, c_void_p]),
    "como_nn_groupnorm_f32": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_float, c_float, c_int, c_void_p]),
    "como_nn_maxpool2_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "como_nn_upsample2x_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "como_nn_normalize_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float),
                                      c_void_p]),
    "como_nn_cov_act_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "como_nn_resize_aa_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "como_nn_resize_aa_f64": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "como_rgb_to_gray_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p]),
    "como_rgb_to_gray_f64": (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p]),
    "como_img_grads_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "como_img_grads_f64": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "como_img_blur_down_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p]),
    "como_img_blur_down_f64": (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p]),
    "como_img_blur_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p]),
    "como_img_blur_f64": (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p]),
    "como_depth_pool2_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "como_depth_pool2_f64": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "como_subselect_pixels_f32": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p] * 3),
    "como_subselect_pixels_f64": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p] * 3),
    "como_track_precalc_jac_f32": (c_int, [c_void_p] * 5 + [c_long, c_void_p]),
    "como_track_precalc_jac_f64": (c_int, [c_void_p] * 5 + [c_long, c_void_p]),
    "como_win_scaffold": (c_int, [ctypes.POINTER(WinArgs), c_void_p]),
    "como_win_priors": (c_int, [ctypes.POINTER(WinArgs), c_void_p]),
    "como_win_logz_ahead_scratch_bytes": (c_long, [c_int, c_int, c_int, c_int]),
    "como_win_logz_ahead": (c_int, [ctypes.POINTER(WinArgs), c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p]),
    "como_win_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_long, c_void_p]),
    "como_se3_normalize_f32": (c_int, [c_void_p, c_int, c_void_p]),
    "como_se3_normalize_f64": (c_int, [c_void_p, c_int, c_void_p]),
    "como_frame_world_f64": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p]),
    "como_track_frame_record_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p, c_void_p]),
    "como_frame_stack_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "como_win_update_checked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p]),
    "como_gram_workspace_bytes": (c_long, []),
    "como_gram_f64": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "como_predictor_f64": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p] * 3),
    "como_se3_inverse_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "como_se3_inverse_f64": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "como_se3_compose_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "como_se3_compose_f64": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"como_amd: {LIB_PATH} is missing -- the HIP extension is not built. "
                "Run `python -m como_amd.build` (or __graft_entry__.build()). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"como_amd: {what} failed with status {rc} "
                           f"({ {1: 'bad argument', 2: 'launch failure'}.get(rc, 'unknown') })")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev_index = {}


def stream_ptr(device=None):
    """The current HIP stream of `device` as an integer handle.  Called before every launch (~25 times per eager GN iteration):
    torch's own raw accessor (0.3 us) instead of `torch.cuda.current_stream(device).cuda_stream` (5.6 us of Python per call)."""
    if _raw_stream is None:
        return torch.cuda.current_stream(device).cuda_stream
    idx = _dev_index.get(device)
    if idx is None:
        d = torch.device("cuda" if device is None else device)
        idx = d.index if d.index is not None else torch.cuda.current_device()
        if device is not None and torch.device(device).index is not None:
            _dev_index[device] = idx                        # (an explicit ordinal never changes; "cuda" follows the current device)
    return _raw_stream(idx)


def capture_graph(fn, device, thread_local=False):
    """Capture `fn()` into a hipGraph (torch.cuda.CUDAGraph).  Returns (graph, fn's result) or (None, error text): a capture
    that an operation inside invalidates leaves the capture stream in capture mode and torch's current stream pointing at it;
    both are undone here (como_abort_capture, stream restored), so the caller can go on launching eagerly.  After one such
    failure every later call returns (None, reason) at once."""
    global _capture_broken
    if _capture_broken:                        # torch's capture machinery does not survive an aborted capture (a second
        return None, _capture_broken           # attempt aborts the process): stay eager for the rest of the process
    prev = torch.cuda.current_stream(device)
    g = torch.cuda.CUDAGraph()
    ctx = torch.cuda.graph(g, capture_error_mode="thread_local" if thread_local else "global")
    try:
        with ctx:
            out = fn()
        return g, out
    except Exception:   # noqa: BLE001
        import traceback
        err = traceback.format_exc()[-1500:]
        cap = getattr(ctx, "capture_stream", None)
        for st in (cap, torch.cuda.current_stream(device)):
            if st is not None:
                lib().como_abort_capture(st.cuda_stream)
        torch.cuda.set_stream(prev)
        try:
            torch.cuda.synchronize(device)
        except Exception:   # noqa: BLE001
            pass
        lib().como_clear_last_error()
        _capture_broken = "graph capture disabled after an earlier capture failed: " + err[-300:]
        return None, err


_capture_broken = ""


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("como_amd: tensors must live on the GPU (the HIP path has no CPU fallback)")


def suffix(dtype):
    if dtype == torch.float16:
        return "f16"
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise RuntimeError(f"como_amd: unsupported dtype {dtype}")
