"""ctypes loader for libppo_hip.so (C ABI: include/ppo_hip.h).

There is no CPU fallback: if the shared library is missing this module raises, and every compute
entry point returns an error status when no HIP device is present (surfaced as PPOError).
"""
import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("PPO_HIP_LIB") or os.path.join(_DIR, "libppo_hip.so")     # PPO_HIP_LIB: A/B builds

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_i8p = C.POINTER(C.c_int8)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)
c_i64p = C.POINTER(C.c_int64)
H = C.c_void_p
HP = C.POINTER(C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64)

# name -> argtypes; every function returns int32 status.  Mirrors include/ppo_hip.h one to one
# (tests/test_abi.py checks the header against this table and against the built library).
SIGNATURES = {
    "ppo_version": [],
    "ppo_last_error": [C.c_char_p, C.c_int64],
    "ppo_device_init": [C.c_int32],
    "ppo_set_stream": [C.c_void_p],
    "ppo_device_synchronize": [],
    "ppo_device_count": [c_i32p],
    "ppo_compute_returns": [c_f32p, c_u8p, C.c_int64, C.c_double, C.c_int32, c_f32p],
    "ppo_compute_returns_tn": [c_f32p, c_u8p, C.c_int64, C.c_int64, C.c_double, C.c_int32, c_f32p],
    "ppo_gae_tn": [c_f32p, c_u8p, c_f32p, C.c_int64, C.c_int64, C.c_double, C.c_double, c_f32p, c_f32p],
    "ppo_categorical_sample": [c_f32p, c_f32p, C.c_int64, C.c_int64, c_i32p, c_f32p, c_i32p],
    "ppo_linear_action_index": [c_i64p, C.c_int64, C.c_int64, c_i64p],
    "ppo_loss_with_entropy": [c_f32p, c_i64p, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_double, c_f64p, c_f64p],
    "ppo_philox4x32_10": [c_u32p, c_u32p, C.c_int64, c_u32p],
    "ppo_env_create": [C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_uint64, HP],
    "ppo_env_destroy": [H],
    "ppo_env_dims": [H, c_i64p, c_i32p, c_i32p, c_i32p],
    "ppo_env_reset": [H],
    "ppo_env_step": [H, c_i32p],
    "ppo_env_get_state": [H, c_i8p, c_u32p],
    "ppo_env_get_reward": [H, c_f32p],
    "ppo_env_get_terminal": [H, c_u8p],
    "ppo_env_get_internal": [H, c_i8p, c_i8p, c_i32p, c_u32p, c_u32p],
    "ppo_set_rollout_persistent": [C.c_int32],
    "ppo_set_rollout_compact": [C.c_int32],
    "ppo_env_check_errors": [H, c_i32p],
    "ppo_env_set_strict_sampling": [H, C.c_int32],
    "ppo_policy_create": [C.c_int32, C.c_int32, C.c_int32, C.c_int32, HP],
    "ppo_policy_set_dtype": [H, C.c_int32],
    "ppo_policy_get_dtype": [H, C.POINTER(C.c_int32)],
    "ppo_policy_destroy": [H],
    "ppo_policy_num_params": [H, c_i64p],
    "ppo_policy_set_params": [H, c_f32p],
    "ppo_policy_get_params": [H, c_f32p],
    "ppo_policy_forward": [H, c_i8p, c_u32p, C.c_int64, C.c_int32, c_f32p],
    "ppo_policy_get_grad": [H, c_f32p],
    "ppo_policy_grad_buffer_dev": [H, HP, c_i64p],
    "ppo_adam_create": [H, C.c_double, C.c_double, C.c_double, C.c_double, HP],
    "ppo_adam_destroy": [H],
    "ppo_adam_get_lr": [H, c_f64p],
    "ppo_adam_set_lr": [H, C.c_double],
    "ppo_adam_get_state": [H, c_f32p, c_f32p, c_f64p],
    "ppo_adam_set_state": [H, c_f32p, c_f32p, c_f64p],
    "ppo_adam_get_epoch_count": [H, c_i64p],
    "ppo_adam_set_epoch_count": [H, C.c_int64],
    "ppo_rollouts_create": [H, C.c_int64, HP],
    "ppo_rollouts_create_shape": [C.c_int64, C.c_int32, C.c_int32, C.c_int64, HP],
    "ppo_rollouts_destroy": [H],
    "ppo_rollouts_len": [H, c_i64p],
    "ppo_rollouts_dims": [H, c_i64p, c_i64p],
    "ppo_collect_rollouts": [H, H, H, C.c_int64, C.c_double, C.c_int32, C.c_int32],
    "ppo_collect_rollouts_episodes": [H, H, H, C.c_int64, C.c_double, C.c_int32],
    "ppo_rollouts_get_states": [H, c_i8p, c_u32p],
    "ppo_rollouts_get_actions": [H, c_i32p],
    "ppo_rollouts_get_probs": [H, c_f32p],
    "ppo_rollouts_get_returns": [H, c_f32p],
    "ppo_rollouts_get_raw_rewards": [H, c_f32p],
    "ppo_rollouts_get_terminal": [H, c_u8p],
    "ppo_rollouts_get_valid": [H, c_u8p],
    "ppo_rollouts_get_full_probs": [H, c_f32p],
    "ppo_rollouts_get_index": [H, c_i64p],
    "ppo_rollouts_set": [H, C.c_int64, c_i8p, c_u32p, c_i32p, c_f32p, c_f32p, c_u8p],
    "ppo_forward_backward": [H, H, c_i64p, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int32],
    "ppo_set_bwd_small_max_tiles": [C.c_int64],
    "ppo_set_bwd_split_bf16": [C.c_int32],
    "ppo_set_train_tile_max_tiles": [C.c_int64],
    "ppo_set_fwd_split_max_states": [C.c_int64],
    "ppo_set_rollout_split_max_envs": [C.c_int64],
    "ppo_adam_apply": [H, H],
    "ppo_last_losses": [H, c_f64p, c_f64p],
    "ppo_step_batch": [H, H, H, c_i64p, C.c_int64, C.c_double, C.c_double, C.c_int32, c_f64p, c_f64p],
    "ppo_train": [H, H, H, C.c_double, C.c_int64, C.c_int32, C.c_double, C.c_int32, c_i64p, C.c_uint64, C.c_int32,
                  C.c_int32, ALLREDUCE_FN, C.c_void_p, c_f64p, c_f64p, c_f64p],
    "ppo_rollouts_attach_disk": [H, C.c_char_p, C.c_int32],
    "ppo_rollouts_detach_disk": [H],
    "ppo_set_disk_async": [C.c_int32],
    "ppo_rollouts_disk_sync": [H],
    "ppo_rollouts_load_disk": [H, C.c_char_p],
    "ppo_average_returns": [H, H, H, C.c_int64, c_f64p, c_f64p],
    "ppo_average_best_returns": [H, H, H, C.c_int64, c_f64p, c_f64p],
    "ppo_average_normalized_returns": [H, H, H, C.c_int64, c_f64p, c_f64p],
    "ppo_evaluate_trajectories": [H, H, H, C.c_int64, C.c_int32, c_f64p],
    "ppo_profile_returns": [C.c_int64, C.c_int64, C.c_double, C.c_int32, c_f64p],
    "ppo_profile_gae": [C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int32, c_f64p],
    "ppo_rollouts_compute_gae": [H, c_f32p, C.c_double, C.c_double, c_f32p, c_f32p],
    "ppo_rccl_probe": [],
    "ppo_rccl_unique_id": [C.c_void_p],
    "ppo_rccl_init": [C.c_int32, C.c_int32, C.c_void_p],
    "ppo_rccl_allreduce": [C.c_void_p, C.c_void_p, C.c_int64],
    "ppo_rccl_comm_info": [c_i32p, c_i32p],
    "ppo_rccl_self_test": [c_i32p],
    "ppo_rccl_finalize": [],
    "ppo_profile_enable": [C.c_int32],
    "ppo_profile_get": [C.c_char_p, c_f64p, c_i64p],
}


class PPOError(RuntimeError):
    """Mirror of the reference's ErrorException / AssertionError (SURVEY.md 8(b) 'Errors')."""

    def __init__(self, status, message):
        super().__init__("[status %d] %s" % (status, message))
        self.status = status


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64; if libppo_hip.so pulled
    in the system runtime first, a later `import torch` would find no GPU (and torch.distributed/RCCL would be
    unusable).  So when torch is installed its bundled runtime is loaded first (same SONAME, so libppo_hip.so
    binds to it) -- torch itself is NOT imported.  PPO_HIP_RUNTIME=system opts out."""
    if os.environ.get("PPO_HIP_RUNTIME", "torch") != "torch":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "libppo_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C proximalpolicyoptimization.jl_amd/csrc`. There is no CPU fallback." % SO_PATH)
        _preload_hip_runtime()
        L = C.CDLL(SO_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here == ABI drift
            fn.argtypes = args
            fn.restype = C.c_int32
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(1024)
    lib().ppo_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(status):
    if status != 0:
        raise PPOError(status, last_error())


def call(name, *args):
    check(getattr(lib(), name)(*args))


def device_count():
    n = C.c_int32(0)
    lib().ppo_device_count(C.byref(n))
    return n.value
