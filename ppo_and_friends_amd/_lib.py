"""
ctypes binding of libppoaf_hip.so (include/ppoaf_hip.h).

There is NO fallback: if the library is missing, cannot be loaded, or a call
returns an error, this module raises.  The product path never computes the hot
path anywhere else.

torch is imported first so that the library's NEEDED libamdhip64.so.7 resolves
to the HIP runtime torch already loaded (same SONAME) -- one runtime per
process, so torch's streams and device pointers are valid in our launches.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPOAF_LIB: a diagnostic build of the same sources (A/B timing of compile-time variants, in-kernel stamps); never a fallback
LIB_PATH = os.environ.get("PPOAF_LIB") or os.path.join(_HERE, "csrc", "libppoaf_hip.so")

MAX_GATHER_FIELDS = 8

_f32p = C.c_void_p
_ptr = C.c_void_p


class GatherField(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p),
                ("row_bytes", C.c_int32), ("_pad", C.c_int32)]


class MlpDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden", C.c_int32), ("depth", C.c_int32),
                ("out_dim", C.c_int32), ("activation", C.c_int32), ("_pad", C.c_int32),
                ("offset", C.c_int64), ("size", C.c_int64), ("log_std_offset", C.c_int64)]


class PpoUpdateArgs(C.Structure):
    """ppoaf_ppo_update_args_t (include/ppoaf_hip.h) -- field order must match the header."""
    _fields_ = [("actor", MlpDesc), ("critic", MlpDesc),
                ("params", C.c_void_p), ("grads", C.c_void_p), ("exp_avg", C.c_void_p),
                ("exp_avg_sq", C.c_void_p), ("slabs", C.c_void_p), ("bucket_total", C.c_int64),
                ("step_counts", C.c_void_p), ("lr", C.c_void_p), ("norm_scratch", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
                ("grad_scale", C.c_float), ("max_norm", C.c_float), ("head_kind", C.c_int32),
                ("obs", C.c_void_p), ("critic_obs", C.c_void_p), ("raw_actions", C.c_void_p),
                ("advantages", C.c_void_p), ("old_log_probs", C.c_void_p),
                ("rewards_to_go", C.c_void_p), ("values", C.c_void_p),
                ("perm", C.c_void_p), ("row_map", C.c_void_p), ("n_rows", C.c_int64),
                ("cursor", C.c_void_p), ("B", C.c_int64), ("batch_stride", C.c_int64),
                ("normalize_values", C.c_int32), ("n_ranks", C.c_int32),
                ("vn_mean", C.c_void_p), ("vn_var", C.c_void_p), ("vn_count", C.c_void_p),
                ("vn_records", C.c_void_p), ("adv_records", C.c_void_p),
                ("normalize_adv", C.c_int32), ("use_huber", C.c_int32),
                ("surr_clip", C.c_float), ("entropy_weight", C.c_float),
                ("kl_loss_weight", C.c_float), ("huber_delta", C.c_float),
                ("min_std", C.c_float), ("inputs_in_batch_order", C.c_int32),
                ("loss_partials", C.c_void_p), ("totals", C.c_void_p),
                ("mb_offset", C.c_int64), ("cursor_advance", C.c_int64),
                ("split_workspace", C.c_void_p), ("split_workspace_bytes", C.c_int64),
                ("xcd_half", C.c_int32), ("row_pairs", C.c_int32)]


ABI_VERSION = 7


class PolicyStepArgs(C.Structure):
    """ppoaf_policy_step_args_t (include/ppoaf_hip.h)."""
    _fields_ = [("actor", MlpDesc), ("critic", MlpDesc), ("params", C.c_void_p),
                ("obs", C.c_void_p), ("critic_obs", C.c_void_p), ("E", C.c_int64),
                ("head_kind", C.c_int32), ("min_std", C.c_float), ("act_lo", C.c_void_p),
                ("act_hi", C.c_void_p), ("forced_raw_action", C.c_void_p),
                ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("normalize_values", C.c_int32), ("_pad", C.c_int32),
                ("vn_mean", C.c_void_p), ("vn_var", C.c_void_p),
                ("raw_action_out", C.c_void_p), ("action_out", C.c_void_p),
                ("logp_out", C.c_void_p), ("value_out", C.c_void_p),
                ("obs_copy_out", C.c_void_p), ("critic_obs_copy_out", C.c_void_p)]


# name -> (restype, argtypes); mirrors include/ppoaf_hip.h one to one.
class IcmUpdateArgs(C.Structure):
    """ppoaf_icm_update_args_t (include/ppoaf_hip.h) -- field order must match the header."""
    _fields_ = [("obs_dim", C.c_int32), ("hidden", C.c_int32), ("action_dim", C.c_int32),
                ("fwd_action_dim", C.c_int32), ("depth_inv", C.c_int32), ("depth_fwd", C.c_int32),
                ("activation", C.c_int32), ("discrete", C.c_int32),
                ("enc_offset", C.c_int64), ("inv_offset", C.c_int64), ("fwd_offset", C.c_int64),
                ("bucket_total", C.c_int64),
                ("params", C.c_void_p), ("grads", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("slabs", C.c_void_p), ("step_count", C.c_void_p), ("lr", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float), ("grad_scale", C.c_float),
                ("obs", C.c_void_p), ("next_obs", C.c_void_p), ("actions", C.c_void_p),
                ("perm", C.c_void_p), ("row_map", C.c_void_p), ("n_rows", C.c_int64),
                ("cursor", C.c_void_p), ("B", C.c_int64), ("batch_stride", C.c_int64),
                ("icm_beta", C.c_float), ("fused_adam", C.c_int32),
                ("act_scratch", C.c_void_p), ("denc_scratch", C.c_void_p), ("loss_partials", C.c_void_p),
                ("totals", C.c_void_p), ("inputs_in_batch_order", C.c_int32), ("_pad", C.c_int32),
                ("split_workspace", C.c_void_p), ("split_workspace_bytes", C.c_int64),
                ("xcd_half", C.c_int32), ("fuse_kernels", C.c_int32)]


class MatUpdateArgs(C.Structure):
    """ppoaf_mat_update_args_t (include/ppoaf_hip.h) -- field order must match the header."""
    _fields_ = [("obs_dim", C.c_int32), ("num_agents", C.c_int32), ("num_actions", C.c_int32), ("embedding", C.c_int32),
                ("offsets", C.c_int64 * 64), ("bucket_total", C.c_int64),
                ("params", C.c_void_p), ("grads", C.c_void_p), ("slabs", C.c_void_p),
                ("critic_obs", C.c_void_p), ("raw_actions", C.c_void_p),
                ("advantages", C.c_void_p), ("old_log_probs", C.c_void_p), ("rewards_to_go", C.c_void_p),
                ("values", C.c_void_p),
                ("perm", C.c_void_p), ("row_map", C.c_void_p), ("n_rows", C.c_int64),
                ("cursor", C.c_void_p), ("B", C.c_int64), ("batch_stride", C.c_int64),
                ("normalize_values", C.c_int32), ("n_ranks", C.c_int32), ("normalize_adv", C.c_int32),
                ("use_huber", C.c_int32),
                ("vn_mean", C.c_void_p), ("vn_var", C.c_void_p), ("vn_count", C.c_void_p),
                ("vn_records", C.c_void_p), ("adv_records", C.c_void_p),
                ("surr_clip", C.c_float), ("entropy_weight", C.c_float), ("kl_loss_weight", C.c_float),
                ("huber_delta", C.c_float),
                ("loss_partials", C.c_void_p), ("totals", C.c_void_p),
                ("norm_scratch", C.c_void_p), ("step_count", C.c_void_p), ("fuse_norm", C.c_int32),
                ("inputs_in_batch_order", C.c_int32),
                ("split_workspace", C.c_void_p), ("split_workspace_bytes", C.c_int64),
                ("mb_offset", C.c_int64), ("cursor_advance", C.c_int64)]


class MatStepArgs(C.Structure):
    """ppoaf_mat_step_args_t (include/ppoaf_hip.h)."""
    _fields_ = [("obs_dim", C.c_int32), ("num_agents", C.c_int32), ("num_actions", C.c_int32), ("embedding", C.c_int32),
                ("actor_obs_dim", C.c_int32), ("normalize_values", C.c_int32),
                ("offsets", C.c_int64 * 64), ("params", C.c_void_p),
                ("critic_obs", C.c_void_p), ("actor_obs", C.c_void_p), ("E", C.c_int64),
                ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("vn_mean", C.c_void_p), ("vn_var", C.c_void_p),
                ("action_out", C.c_void_p), ("raw_action_out", C.c_void_p), ("logp_out", C.c_void_p),
                ("value_out", C.c_void_p), ("critic_obs_copy_out", C.c_void_p), ("obs_copy_out", C.c_void_p),
                ("forced_action", C.c_void_p)]


class ObsFilter(C.Structure):
    """ppoaf_obs_filter_t (include/ppoaf_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("mean", C.c_void_p), ("var", C.c_void_p),
                ("count", C.c_void_p), ("W", C.c_int32), ("normalize", C.c_int32), ("update", C.c_int32),
                ("has_clip", C.c_int32), ("clip_lo", C.c_float), ("clip_hi", C.c_float), ("eps", C.c_float)]


class RewardFilter(C.Structure):
    """ppoaf_reward_filter_t (include/ppoaf_hip.h)."""
    _fields_ = [("reward", C.c_void_p), ("done", C.c_void_p), ("done2", C.c_void_p), ("out", C.c_void_p),
                ("running_reward", C.c_void_p), ("mean", C.c_void_p), ("var", C.c_void_p),
                ("count", C.c_void_p), ("normalize", C.c_int32), ("update", C.c_int32),
                ("has_clip", C.c_int32), ("clip_lo", C.c_float), ("clip_hi", C.c_float),
                ("gamma", C.c_double), ("eps", C.c_double)]


SIGNATURES = {
    "ppoaf_abi_version": (C.c_int, []),
    "ppoaf_last_error": (C.c_char_p, []),
    "ppoaf_device_cu_count": (C.c_int, []),
    "ppoaf_gae_rtg_tmajor": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, C.c_int32, C.c_int64,
                                       C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                       C.c_int, _ptr, _ptr, _ptr]),
    "ppoaf_gae_rtg_tmajor_timed": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, C.c_int32, C.c_int64,
                                             C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                             C.c_int, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_event_create": (C.c_void_p, []),
    "ppoaf_event_destroy": (C.c_int, [_ptr]),
    "ppoaf_event_elapsed_ms": (C.c_int, [_ptr, _ptr, C.POINTER(C.c_float)]),
    "ppoaf_gae_rtg_traj": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, C.c_int64,
                                     C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                     C.c_int, _ptr, _ptr, _ptr]),
    "ppoaf_ppo_loss_fwd_bwd": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, C.c_int64,
                                         C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                         C.c_float, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_minibatch_gather": (C.c_int, [C.POINTER(GatherField), C.c_int32, _ptr, _ptr,
                                         C.c_int64, C.c_int64, _ptr]),
    "ppoaf_scatter_rows_f32": (C.c_int, [_ptr, _ptr, _ptr, C.c_int64, C.c_int64, _ptr, _ptr]),
    "ppoaf_batch_moments": (C.c_int, [_ptr, C.c_int64, C.c_int32, _ptr, _ptr]),
    "ppoaf_running_moments_integrate": (C.c_int, [_ptr, C.c_int32, C.c_int32, _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_normalize": (C.c_int, [_ptr, C.c_int64, C.c_int32, _ptr, _ptr, C.c_float,
                                  C.c_float, C.c_float, C.c_int, _ptr, _ptr]),
    "ppoaf_denormalize": (C.c_int, [_ptr, C.c_int64, C.c_int32, _ptr, _ptr, C.c_float, _ptr, _ptr]),
    "ppoaf_categorical_sample": (C.c_int, [_ptr, C.c_int64, C.c_int32, C.c_uint64, C.c_uint64,
                                           _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_categorical_eval_fwd": (C.c_int, [_ptr, _ptr, C.c_int64, C.c_int32, _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_categorical_eval_bwd": (C.c_int, [_ptr, _ptr, _ptr, _ptr, C.c_int64, C.c_int32, _ptr, _ptr]),
    "ppoaf_gaussian_tanh_eval_fwd": (C.c_int, [_ptr, _ptr, _ptr, C.c_int64, C.c_int32, C.c_float,
                                               _ptr, _ptr, _ptr]),
    "ppoaf_gaussian_tanh_eval_bwd": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, C.c_int64, C.c_int32,
                                               C.c_float, _ptr, _ptr, _ptr]),
    "ppoaf_gaussian_tanh_sample": (C.c_int, [_ptr, _ptr, C.c_int64, C.c_int32, C.c_float, _ptr,
                                             _ptr, C.c_uint64, C.c_uint64, _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_clip_adam_step": (C.c_int, [_ptr, _ptr, _ptr, _ptr, C.c_int64, _ptr, _ptr,
                                       C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       _ptr, _ptr, _ptr]),
    "ppoaf_ppo_update_fwd_bwd": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr]),
    "ppoaf_ppo_update_fwd_bwd_timed": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr, _ptr, _ptr]),
    "ppoaf_ppo_update_reduce": (C.c_int, [C.POINTER(PpoUpdateArgs), C.c_int, _ptr]),
    "ppoaf_ppo_update_split_workspace_bytes": (C.c_int, [C.POINTER(PpoUpdateArgs), C.POINTER(C.c_int64)]),
    "ppoaf_ppo_update_row_pairs_error_offset": (C.c_int, [C.POINTER(PpoUpdateArgs), C.POINTER(C.c_int64)]),
    "ppoaf_ppo_update_split_blocks": (C.c_int, [C.POINTER(PpoUpdateArgs)]),
    "ppoaf_ppo_update_wgrad": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr]),
    "ppoaf_ppo_update_tail_ctl_bytes": (C.c_int, [C.POINTER(PpoUpdateArgs), C.POINTER(C.c_int64)]),
    "ppoaf_ppo_update_wgrad_adam": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr, C.c_double, _ptr]),
    "ppoaf_ppo_update_wgrad_adam_timed": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr, C.c_double, _ptr, _ptr, _ptr]),
    "ppoaf_ppo_update_adam": (C.c_int, [C.POINTER(PpoUpdateArgs), C.c_int, _ptr]),
    "ppoaf_icm_forward_loss_fwd": (C.c_int, [_ptr, _ptr, C.c_int64, C.c_int32, C.c_float, _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_icm_forward_loss_bwd": (C.c_int, [_ptr, _ptr, C.c_int64, C.c_int32, _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_mat_attention_fwd": (C.c_int, [_ptr, _ptr, _ptr, C.c_int64, C.c_int32, C.c_int32, C.c_int, _ptr, _ptr, _ptr]),
    "ppoaf_mat_attention_bwd": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, C.c_int64, C.c_int32, C.c_int32,
                                          _ptr, _ptr, _ptr, _ptr]),
    "ppoaf_policy_step": (C.c_int, [C.POINTER(PolicyStepArgs), _ptr]),
    "ppoaf_minibatch_moments": (C.c_int, [_ptr, _ptr, _ptr, C.c_int64, C.c_int64, _ptr, _ptr]),
    "ppoaf_icm_update_fwd_bwd": (C.c_int, [C.POINTER(IcmUpdateArgs), _ptr]),
    "ppoaf_icm_update_reduce": (C.c_int, [C.POINTER(IcmUpdateArgs), _ptr]),
    "ppoaf_icm_update_split_workspace_bytes": (C.c_int, [C.POINTER(IcmUpdateArgs), C.POINTER(C.c_int64)]),
    "ppoaf_icm_update_fuses_kernels": (C.c_int, [C.POINTER(IcmUpdateArgs)]),
    "ppoaf_icm_intrinsic_reward": (C.c_int, [C.POINTER(IcmUpdateArgs), C.c_float, _ptr, _ptr]),
    "ppoaf_adam_step_prenormed": (C.c_int, [_ptr, _ptr, _ptr, _ptr, C.c_int64, _ptr, _ptr, C.c_float, C.c_float,
                                            C.c_float, C.c_float, C.c_float, _ptr, C.c_int32, _ptr, _ptr]),
    "ppoaf_mat_update_fwd_bwd": (C.c_int, [C.POINTER(MatUpdateArgs), _ptr]),
    "ppoaf_mat_update_fwd_bwd_timed": (C.c_int, [C.POINTER(MatUpdateArgs), _ptr, _ptr, _ptr]),
    "ppoaf_mat_update_reduce": (C.c_int, [C.POINTER(MatUpdateArgs), _ptr]),
    "ppoaf_mat_update_tail_ctl_bytes": (C.c_int, [C.POINTER(MatUpdateArgs), C.POINTER(C.c_int64)]),
    "ppoaf_mat_update_wgrad_adam": (C.c_int, [C.POINTER(MatUpdateArgs), _ptr, _ptr, _ptr, _ptr, C.c_float, C.c_float, C.c_float,
                                              C.c_float, C.c_float, _ptr, C.c_double, _ptr]),
    "ppoaf_mat_update_split_workspace_bytes": (C.c_int, [C.POINTER(MatUpdateArgs), C.POINTER(C.c_int64)]),
    "ppoaf_mat_update_norm_partials": (C.c_int, [C.POINTER(MatUpdateArgs)]),
    "ppoaf_mat_policy_step": (C.c_int, [C.POINTER(MatStepArgs), _ptr]),
    "ppoaf_peer_exchange_create": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "ppoaf_peer_exchange_export": (C.c_int, [_ptr, _ptr]),
    "ppoaf_peer_exchange_connect": (C.c_int, [_ptr, C.c_char_p]),
    "ppoaf_peer_exchange_allreduce": (C.c_int, [_ptr, _ptr, _ptr, C.c_int64, C.c_float, _ptr, C.c_double, _ptr]),
    "ppoaf_peer_exchange_status": (C.c_int, [_ptr, C.POINTER(C.c_int64)]),
    "ppoaf_peer_exchange_destroy": (C.c_int, [_ptr]),
    "ppoaf_comm_unique_id": (C.c_int, [_ptr]),
    "ppoaf_comm_init": (C.c_int, [C.c_int, C.c_int, _ptr, C.POINTER(C.c_void_p)]),
    "ppoaf_allreduce_avg_f32": (C.c_int, [_ptr, _ptr, C.c_int64, _ptr]),
    "ppoaf_bcast_f32": (C.c_int, [_ptr, _ptr, C.c_int64, C.c_int, _ptr]),
    "ppoaf_allreduce_sum_f32": (C.c_int, [_ptr, _ptr, C.c_int64, _ptr]),
    "ppoaf_ppo_update_chain_allreduce": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr, C.c_int64, _ptr]),
    "ppoaf_icm_update_chain_allreduce": (C.c_int, [C.POINTER(IcmUpdateArgs), _ptr, C.c_int64, _ptr, _ptr, _ptr]),
    "ppoaf_mat_update_chain_allreduce": (C.c_int, [C.POINTER(MatUpdateArgs), _ptr, C.c_int64, _ptr, _ptr, _ptr, C.c_float,
                                                   C.c_float, C.c_float, C.c_float, C.c_float, _ptr, _ptr]),
    "ppoaf_allgather_moments": (C.c_int, [_ptr, _ptr, C.c_int64, _ptr, _ptr]),
    "ppoaf_comm_destroy": (C.c_int, [_ptr]),
    "ppoaf_ppo_update_reduce_exchange": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr, C.c_double, _ptr]),
    "ppoaf_ppo_update_wgrad_adam_exchange": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr, C.c_double, _ptr, C.c_double, _ptr]),
    "ppoaf_ppo_update_tail_exchange_floats": (C.c_int, [C.POINTER(PpoUpdateArgs), C.POINTER(C.c_int64)]),
    "ppoaf_ppo_update_adam_exchanged": (C.c_int, [C.POINTER(PpoUpdateArgs), _ptr, _ptr]),
    "ppoaf_env_filter_moments": (C.c_int, [C.POINTER(ObsFilter), C.POINTER(ObsFilter), C.POINTER(RewardFilter),
                                           C.c_int32, C.c_int64, _ptr, _ptr]),
    "ppoaf_env_filter_apply": (C.c_int, [C.POINTER(ObsFilter), C.POINTER(ObsFilter), C.POINTER(RewardFilter),
                                         C.c_int32, C.c_int64, _ptr, C.c_int32, _ptr]),
}

_lib = None


class PpoafError(RuntimeError):
    pass


def _check_single_hip_runtime():
    seen = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    seen.add(os.path.realpath(line.split()[-1]))
    except OSError:
        return
    if len(seen) > 1:
        raise PpoafError(f"two HIP runtimes are mapped in this process: {sorted(seen)}; "
                         "libppoaf_hip.so must share torch's libamdhip64")


def load():
    """Load (once) and type the library.  Raises PpoafError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PpoafError(
            f"{LIB_PATH} is missing: build it with `python -m ppo_and_friends_amd.csrc.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise PpoafError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.ppoaf_abi_version() != ABI_VERSION:
        raise PpoafError(f"ABI version mismatch: library {lib.ppoaf_abi_version()} != {ABI_VERSION}")
    _check_single_hip_runtime()
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ppoaf_last_error().decode("utf-8", "replace")
        raise PpoafError(f"{what or 'libppoaf_hip'} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Requires contiguous CUDA/HIP memory."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PpoafError("libppoaf_hip needs device tensors (got a CPU tensor); no CPU fallback exists")
    if not t.is_contiguous():
        raise PpoafError("libppoaf_hip needs contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream():
    """Current torch HIP stream as a hipStream_t."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
