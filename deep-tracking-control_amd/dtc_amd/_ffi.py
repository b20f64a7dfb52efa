"""ctypes binding of libdtc_hip.so (the C ABI declared in include/dtc_hip.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised.
The library is built in-tree by `deep-tracking-control_amd/build.py` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DTC_LIB") or os.path.join(_HERE, "lib", "libdtc_hip.so")      # DTC_LIB: e.g. the ASan build

c_f32p, c_i64p, c_u8p, c_f64p, c_i32p, c_i16p = (C.c_void_p,) * 6   # device pointers travel as integers
c_stream = C.c_void_p


class DtcError(RuntimeError):
    pass


class DtcGridCfg(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("t_stance", C.c_float), ("fdbk_gain", C.c_float),
                ("x", C.c_float * 64), ("y", C.c_float * 32)]


class DtcSeg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int64), ("col0", C.c_int32), ("width", C.c_int32),
                ("gather", C.c_int32), ("accumulate", C.c_int32), ("rows", C.c_int64), ("amax", C.c_void_p)]


class DtcSegMat(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("cols", C.c_int32), ("idx", C.c_void_p), ("seg", DtcSeg * 4)]


class DtcWgradJob(C.Structure):
    _fields_ = [("dZ", C.c_void_p), ("lddz", C.c_int64), ("X", DtcSegMat), ("dW", C.c_void_p), ("db", C.c_void_p),
                ("N", C.c_int32), ("K", C.c_int32), ("dz_rows", C.c_int64), ("dz_amax", C.c_void_p)]


class DtcFwdLayer(C.Structure):
    _fields_ = [("X", DtcSegMat), ("W", C.c_void_p), ("b", C.c_void_p), ("Y", C.c_void_p), ("ldy", C.c_int64),
                ("N", C.c_int32), ("K", C.c_int32), ("act", C.c_int32)]


class DtcWimgJob(C.Structure):
    _fields_ = [("W", C.c_void_p), ("img", C.c_void_p), ("seg", C.POINTER(DtcSegMat)), ("N", C.c_int32), ("K", C.c_int32),
                ("trans", C.c_int32)]


class DtcH2iWJob(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ld", C.c_int64), ("img", C.c_void_p), ("trans", C.c_int32), ("nrows", C.c_int32), ("nseg", C.c_int32),
                ("r0", C.c_int32 * 2), ("nr", C.c_int32 * 2), ("c0", C.c_int32 * 4), ("cw", C.c_int32 * 4)]


class DtcH2iOperand(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("width", C.c_int32 * 4), ("img", C.c_void_p * 4)]


class DtcH2iFwdLayer(C.Structure):
    _fields_ = [("X", DtcH2iOperand), ("wimg", C.c_void_p), ("b", C.c_void_p), ("Y", C.c_void_p), ("ldy", C.c_int64), ("Yimg", C.c_void_p),
                ("relu_mask", C.c_void_p), ("N", C.c_int32), ("act", C.c_int32)]


class DtcH2iDgradLayer(C.Structure):
    _fields_ = [("dZimg", C.c_void_p), ("wimgT", C.c_void_p), ("dX", C.POINTER(DtcSegMat)), ("dXimg", C.c_void_p), ("add", C.c_void_p),
                ("ld_add", C.c_int64), ("Xsaved", C.c_void_p), ("ldxs", C.c_int64), ("relu_mask", C.c_void_p), ("N", C.c_int32),
                ("Kwin", C.c_int32), ("img_cols", C.c_int32), ("act", C.c_int32)]


class DtcGruFwdItem(C.Structure):
    _fields_ = [("gi", C.c_void_p), ("h0", C.c_void_p), ("W_hh", C.c_void_p), ("b_hh", C.c_void_p), ("hs_all", C.c_void_p),
                ("gates", C.c_void_p), ("hn", C.c_void_p), ("workspace", C.c_void_p)]


class DtcGruBwdItem(C.Structure):
    _fields_ = [("dhs", C.c_void_p), ("hs_all", C.c_void_p), ("gates", C.c_void_p), ("hn", C.c_void_p), ("W_hh", C.c_void_p),
                ("dgi", C.c_void_p), ("dh0", C.c_void_p), ("workspace", C.c_void_p)]


class DtcWgradH2iJob(C.Structure):
    _fields_ = [("dZimg", C.c_void_p), ("Ximg", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("ldw", C.c_int64),
                ("N", C.c_int32), ("K", C.c_int32), ("wcol0", C.c_int32)]


class DtcPpoCfg(C.Structure):
    _fields_ = [("clip_param", C.c_float), ("value_loss_coef", C.c_float), ("entropy_coef", C.c_float),
                ("desired_kl", C.c_float), ("use_clipped_value_loss", C.c_int32),
                ("adaptive_schedule", C.c_int32), ("kl_mirror", C.c_void_p)]


class DtcObsCfg(C.Structure):
    _fields_ = [("ang_vel", C.c_float), ("dof_pos", C.c_float), ("dof_vel", C.c_float),
                ("height_measurements", C.c_float), ("force", C.c_float), ("commands_scale", C.c_float * 3),
                ("base_height_target", C.c_float), ("height_noise", C.c_float), ("term_height", C.c_float),
                ("num_dof", C.c_int32), ("num_foothold_obs", C.c_int32), ("num_points", C.c_int32),
                ("term_row0", C.c_int32), ("term_row1", C.c_int32)]


class DtcEnvStep(C.Structure):
    """include/dtc_hip.h: the buffers of one env step's post-physics block (dtc_env_post_physics); field order as there."""
    _fields_ = ([("height_samples", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("border_size", C.c_float),
                 ("horizontal_scale", C.c_float), ("vertical_scale", C.c_float)] +
                [(k, C.c_void_p) for k in ("root_states", "thigh_pos", "commands", "measured_heights", "idx", "foothold_obs", "opt_world",
                                           "pred", "pred_to_robot", "contact_forces", "termination_contact_indices", "episode_length_buf",
                                           "projected_gravity")] +
                [("max_episode_length", C.c_int64), ("num_bodies", C.c_int32), ("n_term", C.c_int32)] +
                [(k, C.c_void_p) for k in ("reset_buf", "time_out_buf", "height_mean", "foot_positions", "contact_filt", "rew_tracking",
                                           "rew_miss", "base_ang_vel", "dof_pos", "default_dof_pos", "dof_vel", "actions", "forces")] +
                [("ld_forces", C.c_int64)] +
                [(k, C.c_void_p) for k in ("height_noise_offset", "u_obs", "noise_scale_vec", "u_heights", "obs_buf", "privileged_obs_buf",
                                           "heights")])


class DtcRowCopy(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_stride_bytes", C.c_int64), ("width_bytes", C.c_int32)]


class DtcProfRec(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms_total", C.c_double), ("work", C.c_double),
                ("launches", C.c_int64), ("bytes", C.c_double)]


ACT = {None: 0, "none": 0, "relu": 1, "crelu": 1, "elu": 2, "selu": 3, "lrelu": 4, "tanh": 5, "sigmoid": 6}
MAX_OPERAND_ELEMS = (1 << 29) - 1

ABI_VERSION = 15         # DTC_ABI_VERSION of include/dtc_hip.h this binding was written against

_SIGS = {
    "dtc_version": (C.c_int, []),
    "dtc_stream_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "dtc_stream_destroy": (C.c_int, [C.c_void_p]),
    "dtc_abi_sizes": (C.c_int, [C.POINTER(C.c_int64), C.c_int]),
    "dtc_last_error": (C.c_char_p, []),
    "dtc_foothold_plan": (C.c_int, [c_f32p] * 4 + [C.POINTER(DtcGridCfg), c_i64p] + [c_f32p] * 5 +
                          [c_i64p, c_f32p, c_f32p, C.c_int, c_stream]),
    "dtc_foothold_rewards": (C.c_int, [c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, C.c_int, c_stream]),
    "dtc_get_heights": (C.c_int, [c_i16p, C.c_int, C.c_int, c_f32p, C.POINTER(DtcGridCfg), C.c_float, C.c_float,
                                  C.c_float, c_f32p, C.c_int, c_stream]),
    "dtc_foothold_plan_from_table": (C.c_int, [c_i16p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, c_f32p, c_f32p,
                                               c_f32p, C.POINTER(DtcGridCfg), c_f32p, c_i64p] + [c_f32p] * 4 +
                                     [C.c_int, c_stream]),
    "dtc_compute_observations": (C.c_int, [c_f32p] * 11 + [C.c_int64] + [c_f32p] * 4 + [C.POINTER(DtcObsCfg)] +
                                 [c_f32p] * 3 + [C.c_int, c_stream]),
    "dtc_compute_observations_where": (C.c_int, [c_f32p] * 11 + [C.c_int64] + [c_f32p] * 4 + [C.POINTER(DtcObsCfg)] +
                                       [c_f32p] * 3 + [c_u8p, C.c_int, c_stream]),
    "dtc_env_post_physics": (C.c_int, [C.POINTER(DtcEnvStep), C.POINTER(DtcGridCfg), C.POINTER(DtcObsCfg), C.c_int, c_stream]),
    "dtc_check_termination": (C.c_int, [c_f32p, C.c_int, c_i32p, C.c_int, c_i64p, C.c_int64, c_f32p, c_f32p, c_f32p,
                                        C.POINTER(DtcObsCfg), c_u8p, c_u8p, c_f32p, C.c_int, c_stream]),
    "dtc_store_transition": (C.c_int, [C.POINTER(DtcRowCopy), C.c_int, c_f32p, c_f32p, c_u8p, C.c_float, c_f32p,
                                       C.c_int, c_stream]),
    "dtc_history_roll": (C.c_int, [c_f32p, c_f32p, c_f32p, c_u8p, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_gae": (C.c_int, [c_f32p, c_f32p, c_u8p, c_f32p, C.c_float, C.c_float, c_f32p, c_f32p, c_f64p, C.c_int,
                          C.c_int, c_stream]),
    "dtc_adv_sqdev": (C.c_int, [c_f32p, c_f64p, C.c_int64, C.c_double, c_stream]),
    "dtc_adv_normalize": (C.c_int, [c_f32p, c_f64p, C.c_int64, C.c_double, c_stream]),
    "dtc_gather_rows": (C.c_int, [C.c_void_p, c_i64p, C.c_void_p, C.c_int64, C.c_int64, c_stream]),
    "dtc_scatter_rows": (C.c_int, [c_f32p, c_i64p, c_f32p, C.c_int64, C.c_int64, c_stream]),
    "dtc_pack_cols": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, C.c_int64, C.c_int64, C.c_void_p, c_stream]),
    "dtc_linear_fwd": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                 C.c_int, c_stream]),
    "dtc_linear_fwd_list": (C.c_int, [C.POINTER(DtcFwdLayer), C.c_int, C.c_int, c_stream]),
    "dtc_set_gemm_split": (None, [C.c_int]),
    "dtc_get_gemm_split": (C.c_int, []),
    "dtc_s3_planes_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "dtc_s3_wimage_group": (C.c_int, [C.POINTER(DtcWimgJob), C.c_int, c_stream]),
    "dtc_linear_fwd_s3": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_linear_dgrad_s3": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.POINTER(DtcSegMat), c_f32p, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_gru_s3_image_bytes": (C.c_int64, [C.c_int]),
    "dtc_gru_s3_image": (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_int, c_stream]),
    "dtc_gru_step_fwd_s3": (C.c_int, [c_f32p, C.c_void_p] + [c_f32p] * 5 + [C.c_int, C.c_int, c_stream]),
    "dtc_gru_dgrad_parts_s3": (C.c_int, [c_f32p, C.c_void_p, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_probe_mfma_stream": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_f32p, c_stream]),
    "dtc_probe_mfma_stream_h2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_f32p, c_stream]),
    "dtc_probe_poison": (C.c_int, [C.c_uint32, C.c_int, c_f32p, c_stream]),
    "dtc_h2i_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "dtc_h2i_rows64_max": (None, [C.c_int]),
    "dtc_linear_fwd_chain_h2i": (C.c_int, [C.POINTER(DtcH2iFwdLayer), C.c_int, C.c_int, c_stream]),
    "dtc_linear_dgrad_chain_h2i": (C.c_int, [C.POINTER(DtcH2iDgradLayer), C.c_int, C.c_int, c_stream]),
    "dtc_h2i_trace": (None, [C.c_void_p]),
    "dtc_h2i_pack": (C.c_int, [C.POINTER(DtcSegMat), C.c_int, C.c_void_p, c_stream]),
    "dtc_h2i_unpack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_f32p, C.c_int64, c_stream]),
    "dtc_h2i_wimage_bytes": (C.c_int64, [C.POINTER(DtcH2iWJob)]),
    "dtc_h2i_wimage_group": (C.c_int, [C.POINTER(DtcH2iWJob), C.c_int, c_stream]),
    "dtc_linear_fwd_h2i": (C.c_int, [C.POINTER(DtcH2iOperand), C.c_void_p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, c_stream]),
    "dtc_linear_fwd_mse_h2i_parts": (C.c_int64, [C.c_int, C.c_int]),
    "dtc_linear_fwd_mse_h2i": (C.c_int, [C.POINTER(DtcH2iOperand), C.c_void_p, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int, c_i64p, C.c_float,
                                         c_f32p, C.c_int64, C.c_void_p, c_f64p, C.c_int, C.c_int, c_stream]),
    "dtc_linear_dgrad_h2i": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(DtcSegMat), C.c_void_p, C.c_int, c_f32p, C.c_int64, c_f32p,
                                       C.c_int64, C.c_void_p, C.c_int, C.c_int, c_stream]),
    "dtc_wgrad_group_h2i_workspace": (C.c_int64, [C.POINTER(DtcWgradH2iJob), C.c_int, C.c_int]),
    "dtc_wgrad_group_h2i": (C.c_int, [C.POINTER(DtcWgradH2iJob), C.c_int, C.c_int, C.c_void_p, c_stream]),
    "dtc_amax_record_bytes": (C.c_int64, []),
    "dtc_amax": (C.c_int, [C.POINTER(DtcSegMat), C.c_int, C.c_void_p, c_stream]),
    "dtc_h2_wimage_group": (C.c_int, [C.POINTER(DtcWimgJob), C.c_int, c_stream]),
    "dtc_linear_fwd_h2": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_linear_dgrad_h2": (C.c_int, [c_f32p, C.c_int64, C.c_void_p, c_f32p, C.POINTER(DtcSegMat), c_f32p, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_linear_fwd_mse_h2": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int, c_i64p,
                                        C.c_float, c_f32p, C.c_int64, c_f64p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_wgrad_group_h2": (C.c_int, [C.POINTER(DtcWgradJob), C.c_int, C.c_int, C.c_void_p, c_stream]),
    "dtc_linear_fwd_mse_s3_parts": (C.c_int64, [C.c_int, C.c_int]),
    "dtc_linear_fwd_mse_s3": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int, c_i64p,
                                        C.c_float, c_f32p, C.c_int64, c_f64p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_relu_mask_elems": (C.c_int64, [C.c_int, C.c_int]),
    "dtc_linear_fwd_mask": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, c_stream]),
    "dtc_linear_fwd_amax": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_int, c_stream]),
    "dtc_linear_dgrad_mask": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.POINTER(DtcSegMat), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        c_stream]),
    "dtc_linear_dgrad": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.POINTER(DtcSegMat), c_f32p, C.c_int64, C.c_int,
                                   C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_linear_dgrad_split": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                         C.c_int, c_stream]),
    "dtc_linear_wgrad_workspace": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dtc_linear_wgrad": (C.c_int, [c_f32p, C.c_int64, C.POINTER(DtcSegMat), c_f32p, c_f32p, C.c_void_p, C.c_int,
                                   C.c_int, C.c_int, c_stream]),
    "dtc_linear_wgrad_rows": (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_f32p, C.c_int64, C.c_int64, c_i64p, c_f32p, c_f32p, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_wgrad_group_workspace": (C.c_int64, [C.POINTER(DtcWgradJob), C.c_int, C.c_int]),
    "dtc_wgrad_group": (C.c_int, [C.POINTER(DtcWgradJob), C.c_int, C.c_int, C.c_void_p, c_stream]),
    "dtc_wgrad_group_s3_workspace": (C.c_int64, [C.POINTER(DtcWgradJob), C.c_int, C.c_int]),
    "dtc_wgrad_group_s3": (C.c_int, [C.POINTER(DtcWgradJob), C.c_int, C.c_int, C.c_void_p, c_stream]),
    "dtc_cenet_workspace": (C.c_int64, [C.c_int]),
    "dtc_cenet_latent_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_u8p, c_i32p, C.c_void_p, C.c_int, C.c_void_p, c_stream]),
    "dtc_cenet_latent_fwd_img": (C.c_int, [c_f32p, c_f32p, c_f32p, c_u8p, c_i32p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, c_stream]),
    "dtc_cenet_latent_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_i32p, C.c_void_p, C.c_int, C.c_void_p,
                                       c_stream]),
    "dtc_loss_workspace": (C.c_int64, [C.c_int]),
    "dtc_vae_loss": (C.c_int, [c_f32p] * 6 + [c_i64p] + [c_f32p] * 4 + [C.c_void_p, C.c_int, C.c_void_p, c_stream]),
    "dtc_linear_fwd_mse_parts": (C.c_int64, [C.c_int, C.c_int]),
    "dtc_linear_fwd_mse": (C.c_int, [C.POINTER(DtcSegMat), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_int, c_i64p,
                                     C.c_float, c_f32p, C.c_int64, c_f64p, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_vae_loss_fused": (C.c_int, [c_f32p] * 4 + [c_i64p] + [c_f32p] * 2 + [c_f64p, C.c_int, c_f32p, C.c_void_p, C.c_int,
                                                                             C.c_void_p, c_stream]),
    "dtc_vae_loss_fused_img": (C.c_int, [c_f32p] * 4 + [c_i64p] + [c_f32p] * 2 + [c_f64p, C.c_int, c_f32p, C.c_void_p, C.c_int,
                                                                                 C.c_void_p, C.c_void_p, c_stream]),
    "dtc_ppo_loss": (C.c_int, [c_f32p] * 10 + [c_i64p, C.POINTER(DtcPpoCfg)] + [c_f32p] * 4 +
                     [c_f64p, C.c_void_p, C.c_int, C.c_int, c_stream]),
    "dtc_ppo_heads_loss": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int] + [c_f32p] * 4 + [C.c_int] + [c_f32p] * 8 +
                           [c_i64p, C.POINTER(DtcPpoCfg)] + [c_f32p] * 5 + [C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, c_f64p,
                                                                      C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                                      C.c_void_p, c_stream]),
    "dtc_ppo_heads_loss_img": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int] + [c_f32p] * 4 + [C.c_int] + [c_f32p] * 8 +
                               [c_i64p, C.POINTER(DtcPpoCfg)] + [c_f32p] * 5 + [C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, c_f64p,
                                                                          C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8 + [c_stream]),
    "dtc_lr_adapt": (C.c_int, [c_f32p, c_f64p, C.c_float, c_f32p, c_stream]),
    "dtc_gaussian_act": (C.c_int, [c_f32p] * 7 + [C.c_int, C.c_int, c_stream]),
    "dtc_bootstrap_probability": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_stream]),
    "dtc_adam_workspace": (C.c_int64, [C.c_int64]),
    "dtc_clip_adam": (C.c_int, [c_f32p] * 4 + [C.c_int64, C.c_float, c_f64p, C.c_double, C.c_double, C.c_double,
                                               C.c_int64, c_f32p, C.c_void_p, c_stream]),
    "dtc_randn": (C.c_int, [c_f32p, C.c_int64, C.c_uint64, C.c_uint64, c_stream]),
    "dtc_randperm": (C.c_int, [c_i64p, C.c_int64, C.c_uint64, c_stream]),
    "dtc_gru_step_fwd": (C.c_int, [c_f32p] * 7 + [C.c_int, C.c_int, c_stream]),
    "dtc_gru_workspace": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dtc_gru_dgh_offset": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dtc_gru_fwd": (C.c_int, [c_f32p] * 7 + [C.c_void_p, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_gru_bwd": (C.c_int, [c_f32p] * 9 + [C.c_void_p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_set_gru_seq": (None, [C.c_int]),
    "dtc_get_gru_seq": (C.c_int, []),
    "dtc_gru_seq_trace": (None, [C.c_void_p]),
    "dtc_gru_seq_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "dtc_gru_seq_status": (C.c_int, [C.c_int]),
    "dtc_gru_seq_workspace": (C.c_int64, [C.c_int, C.c_int]),
    "dtc_gru_seq_fwd": (C.c_int, [c_f32p] * 7 + [C.c_void_p, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_gru_seq_fwd_pair": (C.c_int, [C.POINTER(c_f32p)] * 7 + [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_gru_fwd_multi": (C.c_int, [C.POINTER(DtcGruFwdItem), C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_gru_bwd_multi": (C.c_int, [C.POINTER(DtcGruBwdItem), C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_set_concurrency_hint": (None, [C.c_int]),
    "dtc_lstm_workspace": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dtc_lstm_fwd": (C.c_int, [c_f32p] * 8 + [C.c_void_p, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_lstm_bwd": (C.c_int, [c_f32p] * 10 + [C.c_void_p, C.c_int, C.c_int, C.c_int, c_stream]),
    "dtc_prof_enable": (None, [C.c_int]),
    "dtc_prof_reset": (None, []),
    "dtc_prof_report": (C.c_int, [C.POINTER(DtcProfRec), C.c_int]),
}

_lib = None


def lib() -> C.CDLL:
    """Load the shared library once; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DtcError(f"{LIB_PATH} not found: run `python deep-tracking-control_amd/build.py` "
                           "(the HIP library is mandatory; there is no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        missing = [name for name in _SIGS if not hasattr(_lib, name)]
        if missing:
            raise DtcError(f"{LIB_PATH} lacks symbols declared in include/dtc_hip.h: {missing}")
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
        _check_abi(_lib)
        _lib.dtc_set_gemm_split(int(os.environ.get("DTC_GEMM_SPLIT", "1") != "0"))      # library-side twin of ops.SPLIT
    return _lib


def _check_abi(l):
    """The loaded library must be the revision this binding describes: same ABI version, same by-value struct layouts
    (DTC_LIB may point at a separately built library, e.g. the ASan build: a stale one would misread every descriptor)."""
    global _lib
    mine = [DtcGridCfg, DtcObsCfg, DtcRowCopy, DtcSeg, DtcSegMat, DtcFwdLayer, DtcWgradJob, DtcPpoCfg, DtcProfRec, DtcWimgJob, DtcH2iWJob,
            DtcH2iOperand, DtcWgradH2iJob, DtcEnvStep, DtcH2iFwdLayer, DtcH2iDgradLayer, DtcGruFwdItem, DtcGruBwdItem]
    sizes = (C.c_int64 * 32)()
    n = l.dtc_abi_sizes(sizes, 32)
    theirs = list(sizes[:n])
    if l.dtc_version() != ABI_VERSION or theirs != [C.sizeof(t) for t in mine]:
        _lib = None
        raise DtcError(f"{LIB_PATH}: ABI mismatch (library version {l.dtc_version()}, struct sizes {theirs}; binding version "
                       f"{ABI_VERSION}, struct sizes {[C.sizeof(t) for t in mine]}): rebuild with deep-tracking-control_amd/build.py")


def exported_symbols():
    return list(_SIGS)


def check(rc: int, what: str):
    if rc != 0:
        raise DtcError(f"{what} failed (rc={rc}): {lib().dtc_last_error().decode()}")


def ptr(t: torch.Tensor | None) -> int | None:
    """Device pointer of a CUDA/HIP tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DtcError("dtc_amd kernels need device tensors (got a CPU tensor)")
    return t.data_ptr()


def cptr(t: torch.Tensor | None, dtype=None) -> int | None:
    """Like ptr() but insists on a contiguous tensor (and optionally a dtype)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise DtcError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise DtcError(f"expected {dtype}, got {t.dtype}")
    return ptr(t)


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def seg(t: torch.Tensor | None, col0: int, width: int, gather: bool = False, accumulate: bool = False,
        ld: int | None = None) -> DtcSeg:
    """Column block [col0, col0+width) of a 2-D row-major tensor `t` (row stride = t.stride(0)).  The block remembers `t` (`_keep`):
    the two-term fp16 path looks the tensor's amax slot up through it (dtc_amd/ops.py: Amax)."""
    s = DtcSeg()
    if t is None:
        s.ptr, s.ld = None, 0
    else:
        assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32
        if t.shape[0] * t.stride(0) > MAX_OPERAND_ELEMS:
            # the GEMM loaders address an operand with 32-bit byte offsets (buffer loads): a source matrix, gathered
            # or not, must stay below 2 GiB -- e.g. privileged observations [T*N, 1389]: T*N <= 386 000 rows per GPU
            raise DtcError(f"operand of {t.shape[0]} x {t.stride(0)} floats exceeds 2^29 elements (2 GiB); shard the rollout")
        s.ptr, s.ld = ptr(t), (t.stride(0) if ld is None else ld)
        s.rows = t.shape[0]
        s._keep = t                    # the descriptor holds a raw pointer: keep the tensor alive with it
    s.col0, s.width, s.gather, s.accumulate = col0, width, int(gather), int(accumulate)
    return s


def segmat(segs, idx: torch.Tensor | None = None) -> DtcSegMat:
    m = DtcSegMat()
    assert 1 <= len(segs) <= 4
    m.nseg = len(segs)
    m.cols = sum(s.width for s in segs)
    m.idx = cptr(idx, torch.int64) if idx is not None else None
    for i, s in enumerate(segs):
        m.seg[i] = s
    # raw pointers inside: tie the lifetime of the tensors (incl. a temporary `idx.to(dev)`) to the descriptor
    m._keep = (idx, [getattr(s, "_keep", None) for s in segs])
    return m


def prof_report():
    n = lib().dtc_prof_report(None, 0)
    arr = (DtcProfRec * max(n, 1))()
    n = lib().dtc_prof_report(arr, n)
    return [dict(name=arr[i].name.decode(), ms_total=arr[i].ms_total, work=arr[i].work,
                 launches=arr[i].launches, bytes=arr[i].bytes) for i in range(n)]
