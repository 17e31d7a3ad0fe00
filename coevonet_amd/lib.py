"""ctypes binding of libcoevo.so (the C ABI in include/coevo.h).

PyTorch-ROCm is only plumbing here: tensors give device memory (``data_ptr()``) and streams.  There is NO CPU or
eager fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COEVO_LIB") or os.path.join(HERE, "libcoevo.so")  # COEVO_LIB: A/B builds

OBS_STRIDE = 12
LOGIT_STRIDE = 8
FC_MAX_ROWS = 32
MPE_STATE_DOUBLES = 24
STAMP_SLOTS = 32
ST_NAMES = {1: "input contains inf or NaN", 2: "output contains inf or NaN after fc1",
            4: "output contains inf or NaN after fc2", 8: "output contains inf or NaN",
            16: "no action selected (current_best_position = -1)",
            32: "persistent rollout: a workgroup timed out waiting for the other rows of its games (COEVO_ST_SYNC_TIMEOUT)"}

DQN_LOGIT_STRIDE = 32
DQN_FC1_TILED = 0x100   # COEVO_DQN_FC1_TILED: or-ed into the channel argument of the layout-dependent DeepQN entry points
DQN_MAX_ROWS = 16
DQN_TASK_DTYPE = np.dtype([("net_off", "<i8"), ("row_begin", "<i4"), ("n_rows", "<i4")])

# numpy view of coevo_fc_task
TASK_DTYPE = np.dtype([("net_off", "<i8"), ("row_begin", "<i4"), ("n_rows", "<i4"), ("D", "<i4"),
                       ("reserved", "<i4")])


class PCG64State(C.Structure):
    _fields_ = [("state_hi", C.c_uint64), ("state_lo", C.c_uint64), ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64)]

    @classmethod
    def from_seed(cls, seed):
        st = np.random.PCG64(seed).state["state"]
        m = (1 << 64) - 1
        return cls(st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m)


class PerturbJob(C.Structure):   # coevo_fc_perturb_job
    _fields_ = [("parent_slab", C.c_void_p), ("parent_idx", C.c_void_p), ("child_slab", C.c_void_p),
                ("sigma_dev", C.c_void_p), ("dist_ref", C.c_void_p), ("dist_partial", C.c_void_p),
                ("child_first", C.c_int32), ("n_children", C.c_int32), ("D", C.c_int32),
                ("stream_lo_first", C.c_uint32), ("stream_hi", C.c_uint32), ("pad", C.c_int32)]


class FinalizeJob(C.Structure):  # coevo_fc_finalize_job
    _fields_ = [("dist_partial", C.c_void_p), ("dist", C.c_void_p), ("head", C.c_void_p), ("n_blocks", C.c_int32),
                ("n", C.c_int32), ("first", C.c_int32), ("pad", C.c_int32)]


class ResetSeg(C.Structure):     # coevo_reset_seg
    _fields_ = [("game_first", C.c_int32), ("count", C.c_int32), ("first_ordinal", C.c_uint64)]


class FinalPack(C.Structure):    # coevo_final_pack
    _fields_ = [("out", C.c_void_p), ("dist", C.c_void_p), ("n_roles", C.c_int32), ("n_local", C.c_int32), ("hof", C.c_int32),
                ("dist_pitch", C.c_int32), ("dist_first", C.c_int32), ("reserved", C.c_int32)]


class RolloutDesc(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("heavy", C.c_void_p), ("n_heavy", C.c_int32), ("heavy_max_rows", C.c_int32),
                ("light", C.c_void_p), ("n_light", C.c_int32), ("light_max_rows", C.c_int32),
                ("state", C.c_void_p), ("n_games", C.c_int32), ("n_cycles", C.c_int32),
                ("row_game", C.c_void_p), ("row_slot", C.c_void_p), ("game_rows", C.c_void_p),
                ("actions", C.c_void_p), ("status", C.c_void_p), ("game_limit", C.c_void_p),
                ("rewards", C.c_void_p), ("pos_first", C.c_int32), ("n_cohorts", C.c_int32),
                ("state_alt", C.c_void_p), ("actions_by_game", C.c_void_p), ("light_stamps", C.c_void_p),
                ("heavy_begin", C.c_void_p), ("light_begin", C.c_void_p),
                ("merged", C.c_int32), ("concurrent_hint", C.c_int32), ("stamps_armed", C.c_int32), ("sync_cleared", C.c_int32),
                ("sync_words", C.c_void_p), ("pack", C.c_void_p)]


class HostCohort(C.Structure):        # coevo_host_cohort
    _fields_ = [("heavy", C.c_void_p), ("light", C.c_void_p), ("games", C.c_void_p), ("n_heavy", C.c_int32),
                ("heavy_max_rows", C.c_int32), ("n_light", C.c_int32), ("light_max_rows", C.c_int32),
                ("n_games", C.c_int32), ("row_first", C.c_int32), ("n_rows", C.c_int32), ("reserved", C.c_int32)]


class HostRolloutDesc(C.Structure):   # coevo_host_rollout_desc
    _fields_ = [("slab", C.c_void_p), ("state", C.c_void_p), ("game_rows", C.c_void_p), ("game_limit", C.c_void_p),
                ("obs_host", C.c_void_p), ("obs_dev", C.c_void_p), ("actions_host", C.c_void_p),
                ("actions_dev", C.c_void_p), ("status", C.c_void_p), ("cohorts", C.c_void_p), ("phase_us", C.c_void_p),
                ("n_games", C.c_int32), ("n_rows", C.c_int32), ("n_cycles", C.c_int32), ("n_cohorts", C.c_int32),
                ("pos_first", C.c_int32), ("zero_copy", C.c_int32), ("reset_ordinals", C.c_void_p),
                ("reset_rng", PCG64State)]


class FrameCohort(C.Structure):       # coevo_frame_cohort
    _fields_ = [("tasks", C.c_void_p * 2), ("rows", C.c_void_p * 2), ("frames_host", C.c_void_p),
                ("frames_dev", C.c_void_p), ("actions_host", C.c_void_p), ("actions_dev", C.c_void_p),
                ("workspace", C.c_void_p), ("n_tasks", C.c_int32 * 2), ("max_rows", C.c_int32 * 2),
                ("game_first", C.c_int32), ("n_games", C.c_int32)]


class FramesRolloutDesc(C.Structure):   # coevo_frames_rollout_desc
    _fields_ = [("slab", C.c_void_p), ("status", C.c_void_p), ("game_state", C.c_void_p), ("acc", C.c_void_p),
                ("game_ordinal0", C.c_void_p), ("limit", C.c_void_p), ("cohorts", C.c_void_p), ("phase_us", C.c_void_p),
                ("generation", C.c_int64), ("ordinals_per_gen", C.c_int64), ("seed", C.c_uint64),
                ("n_games", C.c_int32), ("n_cohorts", C.c_int32), ("C", C.c_int32), ("n_actions", C.c_int32),
                ("T", C.c_int32), ("reserved", C.c_int32)]


class GaSelectRole(C.Structure):
    _fields_ = [("dist", C.c_void_p), ("rewards", C.c_void_p), ("diversity", C.c_void_p), ("fitness", C.c_void_p),
                ("order", C.c_void_p), ("best_dist", C.c_void_p), ("game_first", C.c_int32), ("slot", C.c_int32)]


class GaAdaptArgs(C.Structure):     # coevo_ga_adapt_args
    _fields_ = [("rewards", C.c_void_p), ("gen_dev", C.c_void_p), ("hist", C.c_void_p), ("sig_hist", C.c_void_p),
                ("sigma64", C.c_void_p), ("sigma32", C.c_void_p), ("sigma32_prev", C.c_void_p), ("sig_min", C.c_double),
                ("sig_max", C.c_double), ("eval_first_game", C.c_int32), ("cap", C.c_int32), ("adaptive", C.c_int32),
                ("reserved", C.c_int32)]


class GaPromoteRole(C.Structure):
    _fields_ = [("pop", C.c_void_p), ("hof", C.c_void_p), ("elite", C.c_void_p), ("order", C.c_void_p),
                ("D", C.c_int32), ("elites_from_pop", C.c_int32), ("best_to_pop0", C.c_int32), ("reserved", C.c_int32)]


class CoevoError(RuntimeError):
    pass


_lib = None
_SIGS = {
    "coevo_version": (C.c_int, []),
    "coevo_build_flags": (C.c_char_p, []),
    "coevo_fc_param_count": (C.c_int64, [C.c_int]),
    "coevo_fc_slab_stride": (C.c_int64, [C.c_int]),
    "coevo_fc_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "coevo_fc_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "coevo_fc_forward_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "coevo_fc_forward_merged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "coevo_mpe_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, PCG64State, C.c_uint64, C.c_void_p]),
    "coevo_mpe_reset_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, PCG64State, C.c_void_p]),
    "coevo_mpe_reset_multi_arm": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, PCG64State, C.c_void_p, C.c_int,
                                            C.c_void_p]),
    "coevo_mpe_reset_multi_prep": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, PCG64State, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_ga_select_adapt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "coevo_ga_promote_tick": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_fc_perturb_dist_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_fc_distance_finalize_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_mpe_reset_gen": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, PCG64State, C.c_int64, C.c_void_p,
                                      C.c_int64, C.c_void_p]),
    "coevo_mpe_observe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_mpe_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                 C.c_void_p]),
    "coevo_mpe_host_reset": (C.c_int, [C.c_void_p, C.c_int, PCG64State, C.c_void_p]),
    "coevo_mpe_host_reset_games": (C.c_int, [C.c_void_p, C.c_int, PCG64State, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "coevo_mpe_host_observe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_mpe_host_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "coevo_mpe_host_step_games": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "coevo_host_rollout_create": (C.c_void_p, [C.c_int, C.c_int]),
    "coevo_host_rollout_destroy": (None, [C.c_void_p]),
    "coevo_host_rollout_threads": (C.c_int, [C.c_void_p]),
    "coevo_host_rollout_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "coevo_mpe_host_rollout": (C.c_int, [C.c_void_p, C.POINTER(HostRolloutDesc), C.c_void_p]),
    "coevo_host_placement_choose": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p]),
    "coevo_host_rollout_placement": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_host_rollout_alloc": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "coevo_host_rollout_debug_seed_counters": (C.c_int, [C.c_void_p, C.c_uint32]),
    "coevo_dqn_host_frames_rollout": (C.c_int, [C.c_void_p, C.POINTER(FramesRolloutDesc), C.c_void_p]),
    "coevo_synth_frame_host": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int]),
    "coevo_mpe_rewards": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_mpe_policy_cycle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "coevo_mpe_cycle_kernel_form": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "coevo_mpe_persistent_sync_words": (C.c_int, [C.c_int]),
    "coevo_mpe_persistent_fits": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "coevo_mpe_rollout_persistent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_void_p]),
    "coevo_rollout_ctx_create": (C.c_void_p, [C.c_int]),
    "coevo_rollout_ctx_destroy": (None, [C.c_void_p]),
    "coevo_ga_select": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "coevo_ga_select_gathered": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_ga_promote": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "coevo_ga_promote_rebuild": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32,
                                           C.c_void_p, C.c_void_p]),
    "coevo_mpe_final_step_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "coevo_fc_distance_finalize_multi_tick": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_rollout_ctx_reserve_cohorts": (C.c_int, [C.c_void_p, C.c_int]),
    "coevo_rollout_ctx_cohort_stream": (C.c_void_p, [C.c_void_p, C.c_int]),
    "coevo_rollout_ctx_reset_timing": (C.c_int, [C.c_void_p]),
    "coevo_rollout_ctx_light_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "coevo_mpe_rollout": (C.c_int, [C.POINTER(RolloutDesc), C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_mpe_policy_cycle_stamped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]),
    "coevo_dqn_param_count": (C.c_int64, [C.c_int, C.c_int]),
    "coevo_dqn_slab_stride": (C.c_int64, [C.c_int, C.c_int]),
    "coevo_dqn_workspace_bytes": (C.c_int64, [C.c_int]),
    "coevo_dqn_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "coevo_dqn_forward_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "coevo_dqn_forward_argmax_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int, C.c_void_p]),
    "coevo_dqn_forward_hidden_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_dqn_out_synth_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "coevo_timing_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "coevo_timing_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "coevo_dqn_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "coevo_dqn_relayout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "coevo_dqn_perturb_blocks": (C.c_int64, [C.c_int, C.c_int]),
    "coevo_dqn_perturb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "coevo_dqn_es_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_dqn_es_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_float, C.c_void_p]),
    "coevo_net_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "coevo_synth_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_uint64, C.c_void_p]),
    "coevo_mpe_policy_cycle_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "coevo_mpe_policy_cycle_merged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_int, C.c_void_p]),
    "coevo_mpe_final_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p]),
    "coevo_noise_rounds": (C.c_int, []),
    "coevo_philox4x32": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_philox_normals": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_fc_perturb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    "coevo_fc_perturb_flags": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    "coevo_es_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_es_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                 C.c_float, C.c_void_p]),
    "coevo_centered_ranks": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_fc_rebuild_elites": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64,
                                          C.c_uint32, C.c_void_p, C.c_void_p]),
    "coevo_fc_perturb_gen": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_fc_perturb_blocks": (C.c_int64, [C.c_int]),
    "coevo_fc_perturb_dist": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "coevo_fc_distance_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p]),
    "coevo_gather_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_ga_adapt_sigma": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "coevo_counter_add": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "coevo_fc_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "coevo_es_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                  C.c_void_p]),
    "coevo_fc_diversity": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "coevo_fc_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_sharing_score": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "coevo_ga_fitness": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "coevo_rank_desc": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
}


def exported_symbols():
    return sorted(_SIGS)


def load():
    """dlopen libcoevo.so; raises CoevoError when it has not been built (python -m coevonet_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CoevoError(f"{LIB_PATH} is missing: build the HIP extension first "
                             "(python -m coevonet_amd.build); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        flags = (L.coevo_build_flags() or b"").decode()
        if flags and os.environ.get("COEVO_ALLOW_VARIANT") != "1":
            raise CoevoError(f"{LIB_PATH} was built with non-default switches ({flags}): a measurement variant, not the "
                             "product (rebuild with python -m coevonet_amd.build, or set COEVO_ALLOW_VARIANT=1 in tools/)")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise CoevoError(f"{what} failed with code {rc} "
                         f"({'bad argument' if rc == -1 else 'HIP runtime error' if rc == -2 else 'unsupported here' if rc == -3 else 'unknown'})")


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    _check(getattr(load(), name)(*args, _stream()), name)


def raise_on_status(status_tensor):
    """Turn COEVO_ST_* bits into the ValueError the reference raises (MPE/fcnetwork.py:39-65,87)."""
    st = int(status_tensor.item())
    if st:
        msgs = [m for b, m in ST_NAMES.items() if st & b]
        raise ValueError("\n\t Warning: " + "; ".join(msgs))


def fc_param_count(D):
    return int(load().coevo_fc_param_count(D))


def fc_slab_stride(D):
    return int(load().coevo_fc_slab_stride(D))


def host_tensor(ctx, shape, dtype):
    """zero-filled page-locked, device-mapped host tensor owned by the host-rollout context `ctx` (coevo_host_rollout_alloc:
    first touched on the NUMA node the context's cores run on); valid until the context is destroyed"""
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    np_dtype = np.dtype(dtype)
    nbytes = max(int(np.prod(shape)) * np_dtype.itemsize, 1)
    ptr = load().coevo_host_rollout_alloc(ctx, nbytes)
    if not ptr:
        raise CoevoError("coevo_host_rollout_alloc failed")
    buf = (C.c_char * nbytes).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np_dtype, count=int(np.prod(shape))).reshape(shape))


def host_placement(ctx, max_cpus=256):
    """-> {"cpus": [...], "gpu_numa_node": n, "cpu_numa_node": n, "pinned": bool, "on_gpu_node": bool, "one_l3": bool}"""
    cpus = np.zeros(max_cpus, dtype=np.int32)
    info = np.zeros(4, dtype=np.int32)
    n = load().coevo_host_rollout_placement(ctx, cpus.ctypes.data, max_cpus, info.ctypes.data)
    if n < 0:
        raise CoevoError(f"coevo_host_rollout_placement failed with code {n}")
    return {"cpus": [int(c) for c in cpus[:n]], "gpu_numa_node": int(info[0]), "cpu_numa_node": int(info[1]),
            "pinned": bool(info[3]), "on_gpu_node": bool(info[2] & 1), "one_l3": bool(info[2] & 2),
            "mode": os.environ.get("COEVO_HOST_PIN", "1")}


def tasks_to_device(tasks_np, device="cuda"):
    """numpy structured array (TASK_DTYPE) -> device byte tensor usable as coevo_fc_task*"""
    assert tasks_np.dtype in (TASK_DTYPE, DQN_TASK_DTYPE)
    return torch.from_numpy(tasks_np.view(np.uint8).copy()).to(device)
