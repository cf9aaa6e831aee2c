"""ctypes binding of libarmenv.so (C ABI: include/armenv.h).  No fallback: if the library is missing or
no HIP device is usable, calls raise ``ArmEnvError``."""
import ctypes as C
import os

NJ = 7
ABI_VERSION = 6
TASK_REACH, TASK_PUSH, TASK_PICK = 0, 1, 2
ROBOT_KUKA, ROBOT_DIANA = 0, 1
FK_AUTO, FK_GENERIC = 0, 1
POLICY_EXTERNAL, POLICY_RANDOM, POLICY_ACTOR, POLICY_ACTOR_F16X3, POLICY_DATD3, POLICY_DADDPG = 0, 1, 2, 3, 4, 5

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARMENV_LIB: an alternative build of the same library (A/B timing of two kernel versions inside one GPU session)
LIB_PATH = os.environ.get("ARMENV_LIB") or os.path.join(_HERE, "libarmenv.so")


class ArmEnvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"armenv error {code}: {msg}")
        self.code = code


class ArmEnvChain(C.Structure):
    _fields_ = [("origin_xyz", C.c_double * 3 * NJ), ("origin_rpy", C.c_double * 3 * NJ),
                ("limit_lo", C.c_double * NJ), ("limit_hi", C.c_double * NJ),
                ("base_xyz", C.c_double * 3), ("base_rpy", C.c_double * 3)]


class ArmEnvConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("num_envs", C.c_int64),
        ("task", C.c_int32), ("precision", C.c_int32), ("fk_path", C.c_int32), ("auto_reset", C.c_int32),
        ("seed", C.c_uint64), ("env_id_offset", C.c_uint64),
        ("dv", C.c_double), ("reach_dis", C.c_double), ("max_steps", C.c_int32), ("clamp_joint_limits", C.c_int32),
        ("box_lo", C.c_double * 3), ("box_hi", C.c_double * 3), ("goal_lo", C.c_double * 3), ("goal_hi", C.c_double * 3),
        ("target_quat", C.c_double * 4), ("q_init", C.c_double * NJ),
        ("ik_lambda", C.c_double), ("ik_residual", C.c_double), ("ik_max_dtheta", C.c_double),
        ("ik_max_iters", C.c_int32), ("ik_exit_mode", C.c_int32), ("ik_angle_f32", C.c_int32), ("fence_counters", C.c_int32),
        ("push_success_dis", C.c_double), ("push_cube_half", C.c_double), ("push_eef_radius", C.c_double),
        ("push_rest_z", C.c_double), ("push_place_min", C.c_double), ("push_place_max", C.c_double), ("push_place_z", C.c_double),
        ("pick_gripper_length", C.c_double), ("pick_trigger_dis", C.c_double), ("pick_jaw_half", C.c_double),
        ("fence_z", C.c_double), ("fence_pivot", C.c_double), ("limit_erp", C.c_double), ("ik_tip_offset", C.c_double * 3),
        ("push_tool_radius", C.c_double), ("push_tool_below", C.c_double), ("push_contact_erp", C.c_double),
        ("push_contact_split", C.c_double), ("push_friction", C.c_double), ("push_gravity", C.c_double), ("push_dt", C.c_double),
        ("push_drop_contact", C.c_double), ("push_drop_relax", C.c_double), ("push_contact_model", C.c_int32), ("reserved0", C.c_int32),
        ("rollout_ready_lanes", C.c_int32), ("rollout_waves_per_simd", C.c_int32),
        ("rollout_lanes_per_wave", C.c_int32), ("rollout_straggler_trips", C.c_int32),
        ("chain", ArmEnvChain),
    ]


class ArmEnvMlp(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("W1", "b1", "W2", "b2", "W3", "b3")]


class ArmEnvHerArgs(C.Structure):
    _fields_ = [
        ("T", C.c_int64), ("N", C.c_int64), ("ring_base", C.c_int64), ("ring_cap", C.c_int64),
        ("obs_dim", C.c_int32), ("use_her", C.c_int32),
        ("obs0_dev", C.c_void_p), ("obs_after_dev", C.c_void_p), ("next_obs_dev", C.c_void_p), ("action_dev", C.c_void_p),
        ("reward_dev", C.c_void_p), ("done_dev", C.c_void_p), ("episodes_dev", C.c_void_p), ("num_episodes_dev", C.c_void_p),
        ("batch", C.c_int64), ("picks_dev", C.c_void_p), ("seed", C.c_uint64), ("draw", C.c_uint64),
        ("her_ratio", C.c_float), ("dis_threshold", C.c_float),
        ("states_dev", C.c_void_p), ("actions_dev", C.c_void_p), ("next_states_dev", C.c_void_p), ("rewards_dev", C.c_void_p),
        ("dones_dev", C.c_void_p), ("picks_out_dev", C.c_void_p),
    ]


# every symbol include/armenv.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "armenv_default_config": (C.c_int, [C.c_int32, C.POINTER(ArmEnvConfig)]),
    "armenv_builtin_chain": (C.c_int, [C.c_int32, C.POINTER(ArmEnvChain)]),
    "armenv_create": (C.c_int, [C.POINTER(ArmEnvConfig), C.POINTER(_P)]),
    "armenv_destroy": (None, [_P]),
    "armenv_reset": (C.c_int, [_P, _P, _P, _P]),
    "armenv_reset_with_goal": (C.c_int, [_P, _P, _P, _P, _P]),
    "armenv_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "armenv_rollout": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "armenv_fk": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P]),
    "armenv_ik": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P]),
    "armenv_get_state": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "armenv_set_state": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "armenv_episode_stats": (C.c_int, [_P, _P, _P, _P, _P]),
    "armenv_counters": (C.c_int, [_P, C.POINTER(C.c_uint64 * 16), _P]),
    "armenv_summary": (C.c_int, [_P, _P, _P]),
    "armenv_set_policy": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_float, C.c_float, C.c_float, _P]),
    "armenv_actor_forward": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "armenv_set_policy_datd3": (C.c_int, [_P, C.POINTER(ArmEnvMlp), C.POINTER(ArmEnvMlp), C.POINTER(ArmEnvMlp), C.POINTER(ArmEnvMlp),
                                          C.c_int32, C.c_float, C.c_float, C.c_float, _P]),
    "armenv_set_policy_daddpg": (C.c_int, [_P, C.POINTER(ArmEnvMlp), C.POINTER(ArmEnvMlp), C.POINTER(ArmEnvMlp),
                                           C.c_int32, C.c_float, C.c_float, C.c_float, _P]),
    "armenv_datd3_forward": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "armenv_episode_returns_f32": (C.c_int, [_P, _P, _P]),
    "armenv_count_episodes": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int32, _P, _P]),
    "armenv_write_episodes": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int32, _P, _P, _P, _P]),
    "armenv_her_sample": (C.c_int, [C.c_int32, C.POINTER(ArmEnvHerArgs), _P]),
    "armenv_probe_issue_rate": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "armenv_probe_clock": (C.c_int, [C.c_int32, _P, C.POINTER(C.c_int32), _P]),
    "armenv_num_envs": (C.c_int64, [_P]),
    "armenv_obs_dim": (C.c_int32, [_P]),
    "armenv_aux_dim": (C.c_int32, [_P]),
    "armenv_action_dim": (C.c_int32, [_P]),
    "armenv_kernel_name": (C.c_char_p, [_P]),
    "armenv_last_error": (C.c_char_p, []),
    "armenv_abi_version": (C.c_int32, []),
}

_lib = None


def load():
    """dlopen libarmenv.so (built by `make -C drl-on-robot-arm_amd` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ArmEnvError(-2, f"{LIB_PATH} not found: build it with `make -C drl-on-robot-arm_amd` "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same soname as
        # /opt/rocm's).  Import torch first so that libarmenv.so binds to the runtime torch uses; with
        # the opposite order torch finds the system runtime already loaded and reports no GPUs.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.armenv_abi_version() != ABI_VERSION:
            raise ArmEnvError(-1, "libarmenv.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise ArmEnvError(rc, load().armenv_last_error().decode())


def default_config(task=TASK_REACH) -> ArmEnvConfig:
    cfg = ArmEnvConfig()
    check(load().armenv_default_config(task, C.byref(cfg)))
    return cfg


def chain_struct(chain) -> ArmEnvChain:
    """armenv.urdf.Chain -> ArmEnvChain."""
    s = ArmEnvChain()
    for j in range(NJ):
        s.origin_xyz[j][:] = chain.origin_xyz[j]
        s.origin_rpy[j][:] = chain.origin_rpy[j]
        s.limit_lo[j] = chain.limit_lo[j]
        s.limit_hi[j] = chain.limit_hi[j]
    s.base_xyz[:] = chain.base_xyz
    s.base_rpy[:] = chain.base_rpy
    return s
