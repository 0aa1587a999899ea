"""
ctypes binding of ``libsafelife_hip.so`` (the C-ABI of include/safelife_hip.h).

This is the only way the package reaches the GPU kernels.  There is no CPU
fallback: if the library is missing, or no HIP device is visible when a compute
entry point is called, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAFELIFE_HIP_LIB selects another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("SAFELIFE_HIP_LIB") or os.path.join(_HERE, "libsafelife_hip.so")

SL_MAX_CELLS = 16384
SL_MAX_CHANNELS = 32
SL_E_SHAPE, SL_E_ARG, SL_E_HIP, SL_E_UNSUPPORTED = -1, -2, -3, -4

#: every symbol include/safelife_hip.h declares
EXPORTS = (
    "slhip_abi_version", "slhip_last_error", "slhip_device_count",
    "slhip_advance_board", "slhip_advance_board_each", "slhip_life_occupancy", "slhip_alive_counts", "slhip_execute_actions",
    "slhip_env_prepare", "slhip_pool_baseline", "slhip_pool_write", "slhip_goal_cache_bytes", "slhip_env_reset", "slhip_env_step", "slhip_env_step_slices", "slhip_env_step_range", "slhip_env_rollout",
    "slhip_streams_concurrent", "slhip_streams_order",
    "slhip_env_obs", "slhip_env_step_multi", "slhip_env_reset_multi",
    "slhip_obs_to_policy", "slhip_sample_actions", "slhip_side_effects",
    "slhip_gather_unique_id", "slhip_gather_init", "slhip_gather_window", "slhip_gather_destroy",
    "slhip_gather_window_async", "slhip_gather_done", "slhip_gather_wait_streams",
    "slhip_gather_window_queued",
    "slhip_queues_open", "slhip_queues_open_on", "slhip_queues_stream_shares", "slhip_gather_stream_shares", "slhip_gather_poke", "slhip_queues_mode", "slhip_queues_steps", "slhip_queues_step", "slhip_queues_stage", "slhip_queues_go", "slhip_queues_marker",
    "slhip_queues_wait", "slhip_queues_sync", "slhip_queues_close", "slhip_queues_selftest",
)
QUEUES_RELEASE_FREE = 1
QUEUES_STAGE_MAX = 48
QUEUES_SELFTEST_PLANT, QUEUES_SELFTEST_SWAP = 1, 2
SL_GATHER_ID_BYTES = 128
SL_SE_MAX_KEYS = 24


class SafeLifeHipError(RuntimeError):
    pass


class Pcg64(C.Structure):
    _fields_ = [("state_hi", C.c_uint64), ("state_lo", C.c_uint64),
                ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64)]


_p = C.c_void_p

#: field names of `struct sl_env_batch`, in order
ENV_SCALARS_HEAD = ("B", "H", "W", "E", "time_limit", "exit_points", "n_tables", "auto_reset",
                    "remove_white_goals", "view_h", "view_w", "n_channels")
ENV_STATE_PTRS = ("board", "goals", "exit_locs", "rng", "scalars", "points_table")
ENV_POOL_PTRS = ("pool_board", "pool_goals", "pool_exit_locs", "pool_rng", "pool_scalars")
ENV_POOL_TAIL = ("pool_next",)      # (optional: set by SafeLifeVectorEnv for refreshable pools)
ENV_OUT_PTRS = ("out", "obs", "score_lut")
SL_ABI_VERSION = 13

#: int32 column of each field inside `struct sl_env_scalars` (64 bytes = 16 columns)
SCALAR_COLS = {"agent_row": 0, "agent_col": 1, "num_steps": 2, "old_value": 3, "required_points": 4,
               "initial_points": 5, "table_idx": 6, "level_idx": 7, "episode_idx": 8, "episode_length": 9,
               "episode_reward": 10, "spawn_prob": 11, "goals_static": 12, "is_active": 13,
               "exit_open_at_reset": 14, "loaded": 15}
SCALAR_FLOATS = ("episode_reward", "spawn_prob")
#: `struct sl_level_scalars` (32 bytes = 8 columns)
LEVEL_COLS = {"agent_row": 0, "agent_col": 1, "required_reset": 2, "required_step": 3, "initial_points": 4,
              "table_idx": 5, "spawn_prob": 6}


WRAP_MOVEMENT, WRAP_AS_PENALTY, WRAP_EXIT_BONUS, WRAP_SIDE_EFFECT, WRAP_IGNORE_REWARD_CELLS, WRAP_INACTION = 1, 2, 4, 8, 16, 32
WRAP_MAX_PERIOD = 8


class WrapState(C.Structure):
    """`struct sl_wrap_state` (48 bytes)."""
    _fields_ = [("n_prior", C.c_int32), ("last_side_effect", C.c_int32),
                ("prior", C.c_int16 * (2 * WRAP_MAX_PERIOD)), ("reserved", C.c_int32 * 2)]


class Wrappers(C.Structure):
    """`struct sl_wrappers`."""
    _fields_ = [("flags", C.c_int32), ("move_period", C.c_int32), ("move_table_len", C.c_int32),
                ("reserved", C.c_int32),
                ("move_bonus", C.c_double), ("exit_bonus", C.c_double), ("penalty_coef", C.c_double),
                ("move_table", _p), ("state", _p), ("shaped_reward", _p), ("shaped_reward_t", _p),
                ("pool_baseline", _p), ("inaction_board", _p), ("inaction_rng", _p), ("inaction_rows", _p)]


class EpisodeRecord(C.Structure):
    """`struct sl_episode_record` (32 bytes)."""
    _fields_ = [("env", C.c_int32), ("level", C.c_int32), ("num_steps", C.c_int32), ("episode_idx", C.c_int32),
                ("spawn_prob", C.c_float), ("episode_reward", C.c_float), ("episode_length", C.c_int32),
                ("success", C.c_uint8), ("times_up", C.c_uint8), ("n_cell_types", C.c_uint8), ("reserved", C.c_uint8)]


class PoolRows(C.Structure):
    """``struct sl_pool_rows`` (slhip_pool_write)."""
    _fields_ = [("n", C.c_int32), ("slot", C.c_void_p), ("board", C.c_void_p), ("goals", C.c_void_p),
                ("exit_locs", C.c_void_p), ("rng", C.c_void_p), ("scalars", C.c_void_p), ("next", C.c_void_p),
                ("next_dst", C.c_void_p)]


class EpisodeQueue(C.Structure):
    """`struct sl_episode_queue`."""
    _fields_ = [("capacity", C.c_int32), ("env_base", C.c_int32), ("count", _p), ("records", _p), ("boards", _p)]


class AgentState(C.Structure):
    """struct sl_agent_state (48 bytes = 12 int32 columns)"""
    _fields_ = [("row", C.c_int32), ("col", C.c_int32), ("old_value", C.c_int32), ("required_points", C.c_int32),
                ("initial_points", C.c_int32), ("table_idx", C.c_int32), ("episode_length", C.c_int32),
                ("episode_reward", C.c_float), ("is_active", C.c_int32), ("reserved", C.c_int32 * 3)]


AGENT_COLS = {"row": 0, "col": 1, "old_value": 2, "required_points": 3, "initial_points": 4, "table_idx": 5,
              "episode_length": 6, "episode_reward": 7, "is_active": 8}
LEVEL_AGENT_COLS = {"row": 0, "col": 1, "required_reset": 2, "required_step": 3, "initial_points": 4, "table_idx": 5}
SL_MAX_AGENTS = 8


class MultiAgent(C.Structure):
    """struct sl_multi_agent"""
    _fields_ = [("n_agents", C.c_int32), ("reserved", C.c_int32), ("agents", C.c_void_p), ("pool_agents", C.c_void_p),
                ("out", C.c_void_p), ("obs", C.c_void_p)]


class EnvBatch(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ENV_SCALARS_HEAD]
        + [("channels", C.c_int32 * SL_MAX_CHANNELS), ("spawner_free", C.c_int32), ("stream_salt", C.c_int32)]
        + [(n, _p) for n in ENV_STATE_PTRS]
        + [("L", C.c_int32), ("level_stride", C.c_int32)]
        + [(n, _p) for n in ENV_POOL_PTRS + ENV_POOL_TAIL]
        + [("out", _p), ("obs", _p), ("policy_obs", _p), ("policy_dtype", C.c_int32), ("out_compact", C.c_int32),
           ("score_lut", _p), ("goal_cache", _p), ("pool_ready", _p)]
        + [("wrap", Wrappers), ("finished", EpisodeQueue)]
    )


_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SafeLifeHipError(
                "safelife_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and must be loaded
        # first so that this library binds to the same runtime (same SONAME) instead of pulling a
        # second copy from /opt/rocm, which would then see no device.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.slhip_abi_version.restype = C.c_int
        L.slhip_last_error.restype = C.c_char_p
        L.slhip_device_count.restype = C.c_int
        L.slhip_advance_board.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, _p, C.c_int, _p, _p]
        L.slhip_advance_board_each.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p]
        L.slhip_life_occupancy.argtypes = [_p, _p, C.c_int, C.c_int, C.c_int, _p, C.c_int, _p, _p]
        L.slhip_alive_counts.argtypes = [_p, _p, C.c_int, C.c_int, _p, _p]
        L.slhip_execute_actions.argtypes = [_p, C.c_int, C.c_int, C.c_int, _p, _p, C.c_int, C.c_int,
                                            C.c_int, _p]
        L.slhip_env_prepare.argtypes = [C.POINTER(EnvBatch), _p]
        if hasattr(L, "slhip_pool_baseline"):
            L.slhip_pool_baseline.argtypes = [C.POINTER(EnvBatch), _p]
        if hasattr(L, "slhip_pool_write"):
            L.slhip_pool_write.argtypes = [C.POINTER(EnvBatch), C.POINTER(PoolRows), _p]
        if hasattr(L, "slhip_goal_cache_bytes"):
            L.slhip_goal_cache_bytes.argtypes = [C.POINTER(EnvBatch), C.POINTER(C.c_int)]
            L.slhip_goal_cache_bytes.restype = C.c_size_t
        L.slhip_env_reset.argtypes = [C.POINTER(EnvBatch), _p, _p]
        L.slhip_env_step.argtypes = [C.POINTER(EnvBatch), _p, _p]
        L.slhip_env_step_slices.argtypes = [C.POINTER(EnvBatch), C.c_int, _p, _p, _p]
        if hasattr(L, "slhip_env_step_range"):
            L.slhip_env_step_range.argtypes = [C.POINTER(EnvBatch), C.c_int, C.c_int, _p, _p]
        L.slhip_streams_concurrent.argtypes = [_p, _p, C.POINTER(C.c_int)]
        L.slhip_streams_order.argtypes = [_p, C.c_int, _p, C.c_int]
        L.slhip_env_rollout.argtypes = [C.POINTER(EnvBatch), _p, C.c_int, _p, _p, _p]
        L.slhip_env_obs.argtypes = [C.POINTER(EnvBatch), _p]
        if hasattr(L, "slhip_env_step_multi"):
            L.slhip_env_step_multi.argtypes = [C.POINTER(EnvBatch), C.POINTER(MultiAgent), _p, _p]
            L.slhip_env_reset_multi.argtypes = [C.POINTER(EnvBatch), C.POINTER(MultiAgent), _p, _p]
        if hasattr(L, "slhip_queues_step"):
            L.slhip_queues_open.argtypes = [C.POINTER(EnvBatch), C.c_int, _p, C.c_int, C.POINTER(C.c_void_p)]
            L.slhip_queues_step.argtypes = [C.c_void_p, C.POINTER(EnvBatch), _p, C.c_int]
            if hasattr(L, "slhip_queues_open_on"):
                L.slhip_queues_open_on.argtypes = [C.POINTER(EnvBatch), C.c_int, _p, _p, C.c_int, C.POINTER(C.c_void_p)]
                L.slhip_queues_stream_shares.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_int)]
                L.slhip_gather_stream_shares.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
                L.slhip_gather_poke.argtypes = [C.c_void_p]
            L.slhip_queues_sync.argtypes = [C.c_void_p]
            L.slhip_queues_close.argtypes = [C.c_void_p]
        if hasattr(L, "slhip_queues_steps"):
            L.slhip_queues_mode.argtypes = [C.c_void_p, C.POINTER(C.c_char_p)]
            L.slhip_queues_steps.argtypes = [C.c_void_p, C.POINTER(EnvBatch), _p, C.c_longlong, C.c_longlong, C.c_int, C.c_int]
            if hasattr(L, "slhip_queues_stage"):
                L.slhip_queues_stage.argtypes = L.slhip_queues_steps.argtypes
                L.slhip_queues_go.argtypes = [C.c_void_p]
            L.slhip_queues_marker.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
            L.slhip_queues_wait.argtypes = [C.c_void_p, C.c_longlong]
            L.slhip_queues_selftest.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.slhip_gather_window_queued.argtypes = [_p, _p, _p, C.c_size_t, C.c_void_p, _p, C.POINTER(C.c_longlong)]
        L.slhip_side_effects.argtypes = [C.POINTER(EnvBatch), C.POINTER(EpisodeQueue), C.c_int, C.c_int] + [_p] * 9
        if hasattr(L, "slhip_sample_actions"):
            L.slhip_sample_actions.argtypes = [_p, C.c_int, C.c_int, C.c_ulonglong, C.c_ulonglong, _p, _p]
        L.slhip_obs_to_policy.argtypes = [_p, C.c_int, C.c_int, C.c_int, _p, C.c_int, _p, C.c_int, _p]
        L.slhip_gather_unique_id.argtypes = [_p]
        L.slhip_gather_init.argtypes = [_p, C.c_int, C.c_int, C.POINTER(_p)]
        L.slhip_gather_window.argtypes = [_p, _p, _p, C.c_size_t, _p]
        L.slhip_gather_destroy.argtypes = [_p]
        L.slhip_gather_window_async.argtypes = [_p, _p, _p, C.c_size_t, _p, C.c_int, _p, C.POINTER(C.c_longlong)]
        L.slhip_gather_done.argtypes = [_p, C.c_longlong, C.c_int, C.POINTER(C.c_int)]
        L.slhip_gather_wait_streams.argtypes = [_p, C.c_longlong, _p, C.c_int]
        foreign = bool(os.environ.get("SAFELIFE_HIP_LIB")) and os.environ.get("SAFELIFE_HIP_LIB_ANY_ABI") == "1"
        for name in EXPORTS:
            if not foreign or hasattr(L, name):      # (A/B runs against an older build: tools/ab_libs.sh)
                getattr(L, name)  # AttributeError here means the .so is stale
        if L.slhip_abi_version() != SL_ABI_VERSION and not foreign:
            raise SafeLifeHipError("%s has ABI version %d, this package needs %d: rebuild it"
                                   % (LIB_PATH, L.slhip_abi_version(), SL_ABI_VERSION))
        _lib = L
    return _lib


def check(rc, shape_message=None):
    """Translate a status code into the exception the reference's wrapper would raise."""
    if rc == 0:
        return
    msg = (lib().slhip_last_error() or b"").decode()
    if rc == SL_E_SHAPE:
        raise ValueError(shape_message or msg)
    if rc == SL_E_ARG:
        raise ValueError(msg)
    raise SafeLifeHipError("libsafelife_hip: %s (status %d)" % (msg, rc))


_device = None


def device():
    """The torch device this process computes on (cuda:LOCAL_RANK).  Raises without a GPU."""
    global _device
    if _device is None:
        import torch
        if not torch.cuda.is_available():
            raise SafeLifeHipError(
                "safelife_amd needs a HIP device (torch.cuda.is_available() is False); "
                "there is no CPU fallback.")
        idx = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(idx)
        _device = torch.device("cuda", idx)
    return _device


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
