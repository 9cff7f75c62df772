"""ctypes binding of libagents_amd.so (the C ABI declared in include/agents_amd.h).

The product path has NO fallback: if the HIP library is missing or a kernel returns an error the
call raises.  Torch is used only for device memory (`data_ptr()`) and the current HIP stream.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p)

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# AA_LIB_PATH: a developer override used by tools/ab_lib.sh to time two builds of the library on
# one box (boxes of the pool differ by several percent); the product loads the in-tree library.
LIB_PATH = os.environ.get("AA_LIB_PATH") or os.path.join(_PKG_DIR, "libagents_amd.so")

AA_ACT_NONE, AA_ACT_RELU, AA_ACT_TANH = 0, 1, 2
AA_A_ROW, AA_A_COL, AA_A_PATCH, AA_A_PATCH_U8, AA_A_PATCH_T, AA_A_PATCH_T_U8 = 0, 1, 2, 3, 4, 5
AA_B_ROW, AA_B_COL = 0, 1
AA_LOSS_HUBER, AA_LOSS_SQUARED, AA_LOSS_TARGETS = 0, 1, 2
AA_OBS_U8, AA_OBS_F32 = 0, 1
AA_SAC_STD_EXP, AA_SAC_STD_CLIP_EXP = 0, 1
AA_PPO_NSTATS = 8
AA_PPO_DIST_STATS = 16 + 6 * 256

AA_ERR_INVALID, AA_ERR_RANGE = -22, -34
_ERRORS = {-22: "AA_ERR_INVALID (bad argument)", -34: "AA_ERR_RANGE (size / workspace)",
           -5: "AA_ERR_LAUNCH (HIP launch failure)", -62: "AA_ERR_TIMEOUT (mailbox wait)",
           -95: "AA_ERR_UNSUPPORTED (HIP runtime version / layout not the analysed one)"}
AA_ERR_UNSUPPORTED = -95


class AgentsAmdError(RuntimeError):
    pass


class ConvLayerDesc(Structure):
    _fields_ = [("w", c_void_p), ("bias", c_void_p), ("y", c_void_p), ("KH", c_int32),
                ("KW", c_int32), ("stride", c_int32), ("Cout", c_int32), ("act", c_int32)]


class ConvDxDesc(Structure):
    _fields_ = [("dz", c_void_p), ("w", c_void_p), ("mask_src", c_void_p), ("dx", c_void_p),
                ("n_img", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32),
                ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("Cout", c_int32),
                ("mask_kind", c_int32)]


class PlaneScatter(Structure):
    """aa_plane_scatter (include/agents_amd.h)."""
    _fields_ = [("n", c_int32), ("stride", c_int32 * 4), ("lo", c_int64 * 4), ("hi", c_int64 * 4),
                ("pos", c_void_p * 4), ("planes", c_void_p * 4)]


class GradSlabs(Structure):
    """aa_grad_slabs (include/agents_amd.h)."""
    _fields_ = [("n", c_int32), ("splits", c_int32 * 4), ("mn", c_int32 * 4),
                ("n_tail", c_int32 * 4), ("offset", c_int64 * 4), ("slab", c_void_p * 4)]


class MlpLayout(Structure):
    """aa_mlp_layout."""
    _fields_ = [("n_layers", c_int32), ("dims", c_int32 * 5), ("acts", c_int32 * 4),
                ("k_off", c_int64 * 4), ("b_off", c_int64 * 4)]


class MlpWideFwd(Structure):
    """aa_mlp_wide_fwd (include/agents_amd.h)."""
    _fields_ = [("layout", MlpLayout), ("n_nets", c_int32), ("x_split", c_int32), ("B", c_int64),
                ("params", c_void_p * 4), ("x", c_void_p * 4), ("ldx", c_int64 * 4),
                ("x2", c_void_p * 4), ("ldx2", c_int64 * 4), ("y", (c_void_p * 4) * 4)]


class SacSampleTail(Structure):
    """aa_sac_sample_tail (include/agents_amd.h)."""
    _fields_ = [("net", c_int32), ("A", c_int32), ("std_kind", c_int32), ("act_mean", c_void_p),
                ("act_mag", c_void_p), ("eps_in", c_void_p), ("seed", c_uint64),
                ("call_counter_dev", c_void_p), ("arrival_dev", c_void_p), ("action", c_void_p),
                ("logp", c_void_p), ("save_tanh", c_void_p), ("save_sigma", c_void_p),
                ("save_eps", c_void_p)]


class SacDoutGen(Structure):
    """aa_sac_dout_gen (include/agents_amd.h)."""
    _fields_ = [("kind", c_int32), ("q1", c_void_p), ("q2", c_void_p), ("tq1", c_void_p),
                ("tq2", c_void_p), ("next_logp", c_void_p), ("reward", c_void_p),
                ("discount", c_void_p), ("logp", c_void_p), ("weights", c_void_p),
                ("log_alpha", c_void_p), ("gamma", c_float), ("reward_scale", c_float),
                ("loss_kind", c_int32), ("loss_weight", c_float), ("global_batch", c_float),
                ("loss_out", c_void_p), ("td_target_out", c_void_p), ("dlogp_out", c_void_p),
                ("z", c_void_p), ("A", c_int32), ("std_kind", c_int32), ("act_mag", c_void_p),
                ("save_tanh", c_void_p), ("save_sigma", c_void_p), ("save_eps", c_void_p),
                ("daction", c_void_p), ("ld_daction", c_int64), ("daction2", c_void_p),
                ("ld_daction2", c_int64), ("dlogp", c_void_p)]


AA_SAC_GEN_CRITIC, AA_SAC_GEN_ACTOR, AA_SAC_GEN_HEAD = 1, 2, 3


class PpoPolicyStepDesc(Structure):
    """aa_ppo_policy_step_desc (include/agents_amd.h)."""
    _fields_ = [("x", c_void_p), ("ldx", c_int64), ("B", c_int64),
                ("nrm_mean", c_void_p), ("nrm_var_num", c_void_p), ("nrm_var_den", c_void_p),
                ("nrm_eps", c_float), ("nrm_clip", c_float),
                ("params_a", c_void_p), ("n_layers_a", c_int32), ("dims_a", POINTER(c_int32)),
                ("acts_a", POINTER(c_int32)), ("k_off_a", POINTER(c_int64)),
                ("b_off_a", POINTER(c_int64)),
                ("params_b", c_void_p), ("n_layers_b", c_int32), ("dims_b", POINTER(c_int32)),
                ("acts_b", POINTER(c_int32)), ("k_off_b", POINTER(c_int64)),
                ("b_off_b", POINTER(c_int64)),
                ("value_out", c_void_p),
                ("std_bias", c_void_p), ("act_mean", c_void_p), ("act_mag", c_void_p),
                ("D", c_int32), ("loc", c_void_p), ("scale", c_void_p),
                ("seed", c_uint64), ("call_counter_dev", c_void_p), ("arrival_dev", c_void_p),
                ("clip_lo", c_void_p), ("clip_hi", c_void_p), ("action", c_void_p)]


class MlpWideBwd(Structure):
    """aa_mlp_wide_bwd (include/agents_amd.h)."""
    _fields_ = [("layout", MlpLayout), ("n_nets", c_int32), ("x_split", c_int32), ("B", c_int64),
                ("params", c_void_p * 4), ("x", c_void_p * 4), ("ldx", c_int64 * 4),
                ("x2", c_void_p * 4), ("ldx2", c_int64 * 4), ("y", (c_void_p * 4) * 4),
                ("dout", c_void_p * 4), ("ld_dout", c_int64 * 4), ("dz", (c_void_p * 4) * 4),
                ("dx", c_void_p * 4), ("ld_dx", c_int64 * 4), ("dx_lo", c_int32),
                ("dx_hi", c_int32), ("grads", c_void_p * 4)]


class MlpWideAdam(Structure):
    """aa_mlp_wide_adam (include/agents_amd.h)."""
    _fields_ = [("p", c_void_p * 4), ("m", c_void_p * 4), ("v", c_void_p * 4),
                ("target", c_void_p * 4), ("lr", c_float), ("beta1", c_float), ("beta2", c_float),
                ("eps", c_float), ("tau", c_float), ("step_dev", c_void_p),
                ("arrival_dev", c_void_p)]


class PpoFusedDesc(Structure):
    """aa_ppo_fused_desc (include/agents_amd.h)."""
    _fields_ = [("obs", c_void_p), ("ld_obs", c_int64), ("obs_dim", c_int32), ("D", c_int32),
                ("actions", c_void_p), ("old_loc", c_void_p), ("old_scale", c_void_p),
                ("returns", c_void_p), ("adv", c_void_p), ("old_vpred", c_void_p),
                ("step_type", c_void_p), ("weights", c_void_p), ("N", c_int64),
                ("rows", c_void_p),
                ("nrm_count", c_void_p), ("nrm_avg", c_void_p), ("nrm_m2", c_void_p),
                ("nrm_eps", c_float), ("nrm_clip", c_float),
                ("params", c_void_p), ("total", c_int64), ("head_off", c_int64),
                ("actor", MlpLayout), ("value", MlpLayout),
                ("act_mean", c_void_p), ("act_mag", c_void_p),
                ("clip_eps", c_float), ("value_clip", c_float), ("c_v", c_float),
                ("c_e", c_float), ("denom", c_float), ("logp_clip", c_float),
                ("adv_eps", c_float)]


class GemmDesc(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32),
        ("a_mode", c_int32), ("b_mode", c_int32),
        ("n_img", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32),
        ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("img_pitch", c_int32),
        ("a_div", c_float),
        ("bias", c_void_p), ("act", c_int32),
        ("mask_src", c_void_p), ("ldm", c_int32), ("mask_kind", c_int32),
        ("force_cfg", c_int32), ("force_splits", c_int32),
        ("colsum_out", c_void_p), ("no_dma", c_int32),
    ]


# name -> (restype, argtypes); mirrors include/agents_amd.h one to one.
_SIGNATURES = {
    "aa_abi_version": (c_int, []),
    "aa_rb_scatter_rows": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int,
                                   c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "aa_rb_scatter_rows_count": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64),
                                         c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                         c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "aa_rb_sample_rows": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_uint64,
                                  c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "aa_rb_draw_tf_host": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int64, c_uint64, c_uint64,
                                   c_uint64, c_uint64, c_void_p, c_void_p]),
    "aa_rb_sample_gather": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                    c_int64, c_int64, c_uint64, c_uint64, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "aa_rb_sample_gather_stamped": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64),
                                            c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                            c_int64, c_int64, c_int64, c_uint64, c_uint64,
                                            c_void_p, c_void_p, c_void_p]),
    "aa_rb_gather_rows": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int,
                                  c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "aa_rb_write_rows": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int,
                                 c_void_p, c_int64, c_void_p]),
    "aa_rb_compact_append": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int,
                                     c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                     c_void_p]),
    "aa_rb_compact_take": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int,
                                   c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "aa_random_permutation": (c_int, [c_int64, c_uint64, c_uint64, c_void_p, c_void_p]),
    "aa_rb_range_rows": (c_int, [c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "aa_counter_add": (c_int, [c_void_p, c_int64, c_void_p]),
    "aa_marker": (c_int, [c_int32, c_void_p]),
    "aa_gemm_f32_workspace_bytes": (c_int64, [POINTER(GemmDesc)]),
    "aa_gemm_f32": (c_int, [POINTER(GemmDesc), c_void_p, c_int64, c_void_p]),
    "aa_gemm_f32_slabs": (c_int, [POINTER(GemmDesc), c_void_p, c_int64, POINTER(c_int32),
                                  c_void_p]),
    "aa_conv_pair_supported": (c_int, [c_int32, c_int32, c_int32, c_int32, POINTER(ConvLayerDesc),
                                       POINTER(ConvLayerDesc)]),
    "aa_conv_pair_forward": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                     POINTER(ConvLayerDesc), POINTER(ConvLayerDesc), c_void_p]),
    "aa_conv_pair_x6_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32,
                                                  POINTER(ConvLayerDesc), POINTER(ConvLayerDesc)]),
    "aa_conv_pair_x6_forward": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                        POINTER(ConvLayerDesc), POINTER(ConvLayerDesc), c_void_p,
                                        c_int64, c_void_p]),
    "aa_conv_pair_x6_phase": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                      POINTER(ConvLayerDesc), POINTER(ConvLayerDesc), c_void_p,
                                      c_int64, c_int32, c_void_p]),
    "aa_conv_dx_frame_supported": (c_int, [POINTER(ConvDxDesc)]),
    "aa_conv_dx_frame": (c_int, [POINTER(ConvDxDesc), c_void_p]),
    "aa_conv_dx_frame_x6_workspace_bytes": (c_int64, [POINTER(ConvDxDesc)]),
    "aa_conv_dx_frame_x6": (c_int, [POINTER(ConvDxDesc), c_void_p, c_int64, c_void_p]),
    "aa_conv_dx_frame_x6_phase": (c_int, [POINTER(ConvDxDesc), c_void_p, c_int64, c_int32,
                                          c_void_p]),
    "aa_conv_dw_frame_x6_workspace_bytes": (c_int64, [POINTER(ConvDxDesc)]),
    "aa_conv_dw_frame_x6": (c_int, [POINTER(ConvDxDesc), c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int64, c_void_p]),
    "aa_conv_dw_frame_x6_slabs": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int64,
                                          c_void_p]),
    "aa_conv_dw_frame_x6_reduce": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "aa_dense_small_forward": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int64,
                                       c_int32, c_int32, c_void_p, c_void_p]),
    "aa_dense_small_forward_slabs": (c_int, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_int32,
                                             c_void_p, c_int64, c_void_p, c_void_p, c_int32,
                                             c_int32, c_void_p, c_void_p]),
    "aa_dense_small_backward": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32,
                                        c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "aa_dense_small_dx": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                  c_void_p, c_void_p]),
    "aa_dense_small_dw": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p,
                                  c_void_p, c_void_p]),
    "aa_mlp_small_forward": (c_int, [c_void_p, c_int64, c_void_p, c_int32, POINTER(c_int32),
                                     POINTER(c_int32), POINTER(c_int64), POINTER(c_int64), c_int64,
                                     POINTER(c_void_p), c_void_p]),
    "aa_mlp_small_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "aa_mlp_wide_supported": (c_int, [POINTER(MlpLayout), c_int64]),
    "aa_mlp_wide_forward": (c_int, [POINTER(MlpWideFwd), c_void_p]),
    "aa_mlp_wide_forward_sample": (c_int, [POINTER(MlpWideFwd), POINTER(SacSampleTail), c_void_p]),
    "aa_mlp_wide_forward_sample2": (c_int, [POINTER(MlpWideFwd), POINTER(SacSampleTail),
                                            POINTER(SacSampleTail), c_void_p]),
    "aa_mlp_wide_backward_gen": (c_int, [POINTER(MlpWideBwd), POINTER(SacDoutGen), c_void_p]),
    "aa_mlp_wide_backward_gen_adam": (c_int, [POINTER(MlpWideBwd), POINTER(SacDoutGen),
                                              POINTER(MlpWideAdam), c_void_p]),
    "aa_mlp_wide_dw_adam": (c_int, [POINTER(MlpWideBwd), POINTER(MlpWideAdam), c_void_p]),
    "aa_mlp_wide_backward": (c_int, [POINTER(MlpWideBwd), c_void_p]),
    "aa_mlp_wide_debug_stamps": (c_int, [c_void_p]),
    "aa_mlp_small_backward": (c_int, [c_void_p, c_int64, c_void_p, c_int32, POINTER(c_int32),
                                      POINTER(c_int32), POINTER(c_int64), POINTER(c_int64), c_int64,
                                      POINTER(c_void_p), c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_int64, c_void_p]),
    "aa_colsum_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "aa_colsum_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                              c_void_p]),
    "aa_act_backward": (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "aa_sumsq_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "aa_col2im_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                              c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "aa_dqn_td_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                               c_double, c_double, c_double, c_int32, c_float, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    "aa_dqn_td_loss_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                    c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                    c_int32, c_int32, c_double, c_double, c_double, c_int32,
                                    c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "aa_dqn_loss_head_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int64, c_int32, c_int32, c_double, c_double, c_double,
                                          c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32,
                                          c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "aa_ppo_fused_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "aa_ppo_fused_merge_apply": (c_int32, [c_int32]),
    "aa_ppo_fused_debug_stamps": (c_int, [c_void_p]),
    "aa_ppo_fused_step": (c_int, [POINTER(PpoFusedDesc), c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                  c_void_p, c_int64, c_void_p]),
    "aa_ppo_fused_epoch": (c_int, [POINTER(PpoFusedDesc), c_void_p, c_int32, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float,
                                   c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "aa_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float,
                             c_float, c_float, c_void_p, c_void_p]),
    "aa_rmsprop_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                c_float, c_float, c_float, c_float, c_void_p]),
    "aa_adam_step_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float,
                                    c_float, c_float, c_float, c_void_p, POINTER(PlaneScatter),
                                    c_void_p]),
    "aa_adam_step_counted": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float,
                                     c_float, c_float, c_float, c_void_p, c_void_p,
                                     POINTER(PlaneScatter), c_void_p]),
    "aa_adam_step_counted_target": (c_int, [c_void_p] * 4 + [c_int64] + [c_float] * 4 +
                                    [c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "aa_rmsprop_step_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_float, c_float, c_float, c_float, POINTER(PlaneScatter),
                                       c_void_p]),
    "aa_rmsprop_step_slabs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_float, c_float, c_float, c_float, POINTER(PlaneScatter),
                                      POINTER(GradSlabs), c_void_p]),
    "aa_rmsprop_step_slabs_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int64, c_float, c_float, c_float, c_float,
                                           POINTER(PlaneScatter), POINTER(GradSlabs), c_void_p,
                                           c_int32, c_void_p, c_void_p]),
    "aa_sgd_step": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "aa_soft_update": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "aa_segment_sumsq": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "aa_clip_by_norm": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_float, c_int32,
                                c_void_p]),
    "aa_count_steps": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "aa_mailbox_create": (c_int, [c_int64, POINTER(c_void_p), POINTER(c_void_p)]),
    "aa_mailbox_destroy": (c_int, [c_void_p]),
    "aa_mailbox_wait": (c_int, [c_void_p, c_int64, c_int64, POINTER(c_int64)]),
    "aa_eps_greedy_action": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p,
                                     c_uint64, c_void_p, c_void_p, c_int64, c_void_p, c_int32,
                                     c_void_p]),
    "aa_hip_graph_exec_spread": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    "aa_hip_graph_instantiate": (c_int, [c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int32),
                                         POINTER(c_int32), POINTER(c_int32)]),
    "aa_hip_graph_launch": (c_int, [c_void_p, c_void_p]),
    "aa_hip_graph_exec_destroy": (c_int, [c_void_p]),
    "aa_boltzmann_action": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p,
                                    c_uint64, c_void_p, c_void_p, c_int64, c_void_p, c_int32,
                                    c_void_p, c_int32, c_void_p]),
    "aa_vecenv_random_step": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_float, c_float,
                                      c_float, c_uint64, c_void_p, c_void_p, c_int32, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "aa_discounted_return": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                     c_int64, c_void_p, c_void_p]),
    "aa_gae": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int64, c_int64,
                       c_int64, c_void_p, c_void_p]),
    "aa_normalize_moments": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "aa_ppo_loss": (c_int, [c_void_p] * 11 + [c_int64, c_int32] + [c_float] * 6 + [c_int32] +
                    [c_void_p] * 5),
    "aa_pack_small_f32": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "aa_pack_sum3_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "aa_copy_segments": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64),
                                 POINTER(c_int64), POINTER(c_int32), c_int32, c_int64, c_void_p]),
    "aa_add_strided_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                   c_void_p]),
    "aa_add_l2_grad": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "aa_ppo_loss_dist": (c_int, [c_void_p] * 11 + [c_int64, c_int32] + [c_float] * 6 +
                         [c_void_p, c_float, c_float] + [c_void_p] * 5),
    "aa_ppo_head_forward": (c_int, [c_void_p] * 4 + [c_int64, c_int32] + [c_void_p] * 3),
    "aa_ppo_policy_step": (c_int, [POINTER(PpoPolicyStepDesc), c_void_p]),
    "aa_ppo_head_forward_sample": (c_int, [c_void_p] * 4 + [c_int64, c_int32, c_void_p, c_void_p,
                                           c_uint64] + [c_void_p] * 6),
    "aa_ppo_head_backward": (c_int, [c_void_p] * 5 + [c_int64, c_int32] + [c_void_p] * 3),
    "aa_normal_log_prob": (c_int, [c_void_p] * 3 + [c_int64, c_int32, c_void_p, c_void_p]),
    "aa_normal_sample": (c_int, [c_void_p, c_void_p, c_int64, c_uint64, c_void_p, c_void_p,
                                 c_void_p]),
    "aa_uniform_sample": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_uint64, c_void_p,
                                  c_void_p, c_void_p]),
    "aa_ppo_discounts": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_int64, c_void_p,
                                 c_void_p]),
    "aa_ppo_trajectory_mask": (c_int, [c_void_p] * 4 + [c_int64, c_void_p, c_void_p]),
    "aa_ppo_update_kl_beta": (c_int, [c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "aa_prio_workspace_bytes": (c_int64, [c_int64]),
    "aa_prio_sample_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                    c_uint64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "aa_prio_draw_workspace_bytes": (c_int64, [c_int64]),
    "aa_prio_draw_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                  c_uint64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "aa_prio_set": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_int64, c_void_p,
                            c_void_p, c_void_p]),
    "aa_prio_on_add": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "aa_sac_sample": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                              c_uint64] + [c_void_p] * 8),
    "aa_sac_head_backward": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32] +
                             [c_void_p] * 4 + [c_int64, c_void_p, c_int64] + [c_void_p] * 3),
    "aa_sac_critic_loss": (c_int, [c_void_p] * 9 + [c_float, c_float, c_int32, c_float, c_int64,
                                                    c_float] + [c_void_p] * 5),
    "aa_sac_actor_loss": (c_int, [c_void_p] * 5 + [c_float, c_int64, c_float] + [c_void_p] * 5),
    "aa_norm_scratch_floats": (c_int64, [c_int64]),
    "aa_streaming_norm_update": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "aa_ema_norm_update": (c_int, [c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p,
                                   c_void_p]),
    "aa_norm_apply": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_float,
                              c_float, c_void_p, c_void_p]),
    "aa_sac_alpha_loss": (c_int, [c_void_p] * 3 + [c_float, c_int32, c_float, c_int64, c_float] +
                          [c_void_p] * 3),
    "aa_sac_alpha_step": (c_int, [c_void_p] * 3 + [c_float, c_int32, c_float, c_int64, c_float] +
                          [c_void_p] * 5 + [c_float] * 4 + [c_void_p] * 4),
}

_lib = None


def exported_symbols():
    """Names every entry point declared in include/agents_amd.h must resolve to."""
    return sorted(_SIGNATURES)


def load():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AgentsAmdError(
            f"{LIB_PATH} is missing: build it with `python -m agents_amd._build` "
            "(there is no CPU / torch fallback on the product path).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.aa_abi_version() != 21:
        raise AgentsAmdError("libagents_amd.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise AgentsAmdError(f"{what} failed: {_ERRORS.get(rc, rc)}")


import os as _os
# The current stream's handle from torch's C entry points instead of torch.cuda.current_stream()
# (saves ~6-10 us of host time per DQN iteration; -1.8 % on the iteration, round 5).  Default ON
# since round 6; AA_RAW_STREAM=0 is the A/B knob.  History: with this on, round 5's full GPU suite
# died in hipGraphLaunch at its 671st test and the path was made opt-in; the dead agents of earlier
# tests were then only released by the cyclic collector, which the slower path happened to trigger
# more often.  Graphs are now released by reference count and destroyed on an idle device
# (utils/graph.py: _GRAVEYARD; tests/test_gpu_lifetime.py).
_RAW_OK = _os.environ.get("AA_RAW_STREAM", "1") != "0"
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) if _RAW_OK else None
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device, as an int.  Through the two
    C entry points `torch.cuda.current_stream()` itself ends in (that call builds a Stream object
    and resolves the device index in Python: 6-9 us, twice per iteration of the DQN loop)."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise AgentsAmdError(
                "agents_amd kernels need device tensors (torch 'cuda' = HIP on ROCm); got a "
                f"{t.device} tensor.  There is no CPU fallback on the product path.")
