"""
Drop-in counterpart of the reference's native module ``safelife.speedups``
(safelife/speedups_src/module.c:515-556) for the step hot path, running on
MI355X through ``libsafelife_hip.so``.

Same function names, argument meaning, dtypes and error behaviour as the
reference's Python-level functions:

* ``advance_board(board, spawn_prob=0.3, n_step=1)``         module.c:20-49
* ``life_occupancy(board, spawn_prob=0.3, n_step=1000)``     module.c:52-81
* ``alive_counts(board, goals)``                             module.c:99-132
* ``execute_actions(board, locations, actions)``             module.c:155-202
* ``set_bit_generator(bitgen)`` / ``seed(n)``                random.c:25-74

plus ``*_batch`` variants that take device tensors ``[B,H,W]`` and never leave
HBM.  Host arrays are staged through torch tensors (buffer holders only).

Random numbers: the reference draws from one process-global numpy
BitGenerator.  Here the generator's PCG64 state is copied to the device, the
kernel consumes draws in the reference's order, and the advanced state is
written back into the generator, so results *and* generator state afterwards
are identical to the reference.  Only PCG64 (numpy's default) is supported.

Not provided (out of the hot path): gen_pattern,
wrapped_label, render_board.
"""
import ctypes as C

import numpy as np

from . import _hip

NEW_CELL_MASK = 1          # module.c:580-582 (only procgen uses them)
CAN_OSCILLATE_MASK = 2
INCLUDE_VIOLATIONS_MASK = 4

_MASK64 = (1 << 64) - 1


class BoardShapeError(ValueError, SystemError):
    """Raised where the reference returns NULL without an exception (module.c:32-35), which
    CPython reports as SystemError; also a ValueError for callers that expect one."""


# --------------------------------------------------------------------------- RNG bridge

_bit_generator = None


def set_bit_generator(bitgen):
    """Use `bitgen` (an object with the numpy BitGenerator interface) for all further draws.

    Mirrors random.c:25-39; the generator is process-global exactly as in the reference.
    """
    global _bit_generator
    if not hasattr(bitgen, "capsule"):
        raise AttributeError("'%s' object has no attribute 'capsule'" % type(bitgen).__name__)
    _bit_generator = bitgen


def seed(n=0):
    """random.c:41-74: install ``numpy.random.default_rng(n)`` (n == 0: OS entropy)."""
    global _bit_generator
    gen = np.random.default_rng(int(n)) if n else np.random.default_rng()
    _bit_generator = gen.bit_generator


def _current_bitgen():
    if _bit_generator is None:
        seed(0)                      # random.c:76-81
    return _bit_generator


def pcg64_words(bitgen):
    """numpy PCG64 state -> uint64[4] (state_hi, state_lo, inc_hi, inc_lo)."""
    st = bitgen.state
    if st.get("bit_generator") != "PCG64":
        raise TypeError(
            "safelife_amd.speedups supports numpy's PCG64 bit generator only (got %r); the device "
            "kernels restate PCG64's recurrence to stay stream-identical with the reference"
            % (st.get("bit_generator"),))
    s, i = st["state"]["state"], st["state"]["inc"]
    return np.array([s >> 64, s & _MASK64, i >> 64, i & _MASK64], dtype=np.uint64)


def pcg64_set_words(bitgen, words):
    st = bitgen.state
    w = [int(x) for x in words]
    st["state"]["state"] = (w[0] << 64) | w[1]
    st["state"]["inc"] = (w[2] << 64) | w[3]
    bitgen.state = st


# --------------------------------------------------------------------------- staging helpers

def _torch():
    import torch
    return torch


def _to_device(arr, np_dtype):
    """Host ndarray -> device tensor holding the same bytes (uint16/uint64 travel as int16/int64)."""
    torch = _torch()
    a = np.ascontiguousarray(arr, dtype=np_dtype)
    view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint64): np.int64}.get(a.dtype)
    if view is not None:
        a = a.view(view)
    return torch.from_numpy(a).to(_hip.device(), non_blocking=False)


def _to_host(t, np_dtype):
    a = t.cpu().numpy()
    return a.view(np_dtype) if a.dtype != np.dtype(np_dtype) else a


def _coerce_board(board):
    """module.c:28-35: force-cast to C-contiguous uint16, must be 2-d and non-empty."""
    b = np.ascontiguousarray(np.asarray(board).astype(np.uint16, copy=False))
    if b.ndim != 2 or b.size == 0:
        raise BoardShapeError("board must be a non-empty 2-dimensional array")
    return b


# --------------------------------------------------------------------------- batched (device) API

def advance_board_batch(boards, spawn_prob, rng, n_step=1, out=None):
    """boards: int16/uint16 tensor [B,H,W] on the device; spawn_prob: float32 [B];
    rng: int64 [B,4] PCG64 words, advanced in place; n_step: int, or an int32 tensor [B] of per-board
    step counts.  Returns `out` (may be `boards`)."""
    torch = _torch()
    B, H, W = boards.shape
    if out is None:
        out = torch.empty_like(boards)
    if isinstance(n_step, torch.Tensor):        # one step count per board: int32 [B] on the device
        n = n_step.to(device=boards.device, dtype=torch.int32).contiguous()
        if tuple(n.shape) != (B,):
            raise ValueError("n_step tensor must have shape [B]")
        rc = _hip.lib().slhip_advance_board_each(_hip.ptr(boards), _hip.ptr(out), B, H, W, _hip.ptr(spawn_prob),
                                                 _hip.ptr(n), _hip.ptr(rng), _hip.current_stream_ptr())
    else:
        rc = _hip.lib().slhip_advance_board(_hip.ptr(boards), _hip.ptr(out), B, H, W, _hip.ptr(spawn_prob),
                                            int(n_step), _hip.ptr(rng), _hip.current_stream_ptr())
    _hip.check(rc, "Board must be at least 3x3.")
    return out


def life_occupancy_batch(boards, spawn_prob, rng, n_step=1000, out=None):
    """int32 [B,H,W,8] occupancy counts of `n_step` steps per board (advance_board.c:153-189); `rng` advanced in place.
    ``out``: a caller-owned int32 tensor [B,H,W,8] to write into (every element is written) instead of a new one --
    at 4096 boards of 64x64 the result is 537 MB, and a fresh allocation of that size is not free."""
    torch = _torch()
    B, H, W = boards.shape
    if out is None:
        counts = torch.empty((B, H, W, 8), dtype=torch.int32, device=boards.device)
    else:
        counts = out
        if (counts.dtype != torch.int32 or tuple(counts.shape) != (B, H, W, 8) or not counts.is_contiguous()
                or counts.device != boards.device):
            raise ValueError("out must be a contiguous int32 tensor [B,H,W,8] on the boards' device")
    rc = _hip.lib().slhip_life_occupancy(_hip.ptr(boards), _hip.ptr(counts), B, H, W, _hip.ptr(spawn_prob),
                                         int(n_step), _hip.ptr(rng), _hip.current_stream_ptr())
    _hip.check(rc, "Board must be at least 3x3.")
    return counts


def alive_counts_batch(boards, goals):
    torch = _torch()
    if boards.shape != goals.shape:
        raise ValueError("Board and goals must have same size.")
    B = boards.shape[0]
    hw = boards[0].numel() if B else 0
    out = torch.empty((B, 8, 9), dtype=torch.int64, device=boards.device)
    rc = _hip.lib().slhip_alive_counts(_hip.ptr(boards), _hip.ptr(goals), B, hw, _hip.ptr(out),
                                       _hip.current_stream_ptr())
    _hip.check(rc)
    return out


def execute_actions_batch(boards, locs, actions):
    """In place.  boards [B,H,W]; locs int64 [B,A,2] (row, col); actions int64 [B,A] or [B,1]."""
    B, H, W = boards.shape
    A = locs.shape[1]
    if actions.shape[-1] == A and actions.dim() == 2:
        stride, bstride = 1, A
    elif actions.numel() == B:
        stride, bstride = 0, 1
    else:
        raise ValueError("Locations should be shape (n_agent, 2).")
    rc = _hip.lib().slhip_execute_actions(_hip.ptr(boards), B, H, W, _hip.ptr(locs), _hip.ptr(actions),
                                          A, stride, bstride, _hip.current_stream_ptr())
    _hip.check(rc, "Board must be at least 3x3.")


# --------------------------------------------------------------------------- reference-shaped API

def _run_with_global_rng(fn, board, spawn_prob):
    torch = _torch()
    bitgen = _current_bitgen()
    words = pcg64_words(bitgen)
    d_board = _to_device(board[None], np.uint16)
    d_rng = _to_device(words[None], np.uint64)
    d_p = torch.full((1,), float(np.float32(spawn_prob)), dtype=torch.float32, device=d_board.device)
    result = fn(d_board, d_p, d_rng)
    pcg64_set_words(bitgen, _to_host(d_rng, np.uint64)[0])
    return result


def advance_board(board, spawn_prob=0.3, n_step=1):
    """Advance `board` by `n_step` steps; returns a new uint16 array, input untouched."""
    b = _coerce_board(board)
    out = _run_with_global_rng(
        lambda d_b, d_p, d_rng: advance_board_batch(d_b, d_p, d_rng, n_step), b, spawn_prob)
    return _to_host(out, np.uint16)[0]


def life_occupancy(board, spawn_prob=0.3, n_step=1000):
    """int32 [H,W,8]: how often each cell held life of each colour over `n_step` steps."""
    b = _coerce_board(board)
    out = _run_with_global_rng(
        lambda d_b, d_p, d_rng: life_occupancy_batch(d_b, d_p, d_rng, n_step), b, spawn_prob)
    return _to_host(out, np.int32)[0]


def alive_counts(board, goals):
    """int64 [8,9]: rows = goal colour, columns = cell colour (k r g y b m c w) + empty."""
    b = np.ascontiguousarray(np.asarray(board).astype(np.uint16, copy=False))
    g = np.ascontiguousarray(np.asarray(goals).astype(np.uint16, copy=False))
    if b.size != g.size:
        raise ValueError("Board and goals must have same size.")
    out = alive_counts_batch(_to_device(b.reshape(1, -1), np.uint16), _to_device(g.reshape(1, -1), np.uint16))
    return _to_host(out, np.int64)[0]


def execute_actions(board, locations, actions):
    """Perform an action for each agent, in order; `board` and `locations` are updated in place."""
    if not isinstance(board, np.ndarray) or not isinstance(locations, np.ndarray):
        raise TypeError("board and locations must be numpy arrays (they are updated in place)")
    if board.ndim != 2:
        raise ValueError("Board should be 2-dimensional.")
    if board.shape[0] < 3 or board.shape[1] < 3:
        raise ValueError("Board must be at least 3x3.")
    acts = np.ascontiguousarray(np.asarray(actions), dtype=np.int64).reshape(-1)
    n_agents = locations.size // 2
    if acts.size != n_agents and acts.size != 1:
        raise ValueError("Locations should be shape (n_agent, 2).")
    if n_agents == 0:
        return None
    d_board = _to_device(board[None], np.uint16)
    d_locs = _to_device(locations.reshape(1, n_agents, 2), np.int64)
    d_acts = _to_device(acts.reshape(1, -1), np.int64)
    B, H, W = d_board.shape
    stride = 1 if acts.size == n_agents else 0
    rc = _hip.lib().slhip_execute_actions(_hip.ptr(d_board), 1, H, W, _hip.ptr(d_locs), _hip.ptr(d_acts),
                                          n_agents, stride, acts.size, _hip.current_stream_ptr())
    _hip.check(rc, "Board must be at least 3x3.")
    np.copyto(board, _to_host(d_board, np.uint16)[0], casting="unsafe")
    np.copyto(locations, _to_host(d_locs, np.int64)[0].reshape(locations.shape), casting="unsafe")
    return None
