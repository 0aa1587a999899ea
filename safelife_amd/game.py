"""
One SafeLife game on the host, stepped through the GPU ``speedups`` functions (compat tier).

This is NOT the reference's class hierarchy re-typed: it is a small state holder written against the
interface that ``safelife_amd.env.SafeLifeEnv``, the reward wrappers and an episode logger use
(SURVEY.md section 7), with the reward glue derived from the rules in SURVEY.md Appendix C and checked
against the reference's recorded traces (tests/test_hip_parity.py::test_compat_env_trace).  What it
offers, under the reference's names so that reference-style drivers run unchanged:

    state        board, goals, agent_locs, agent_names, num_steps, game_over
    constants    spawn_prob, min_performance, points_table, points_on_level_exit, exit_locs,
                 file_name, title, seed / rng
    physics      execute_actions(actions), advance_board(), update_exit_colors()
    scoring      alive_counts, current_points(), points_earned(), initial_available_points(),
                 required_points(), can_exit(), has_exited(), agent_is_active()
    lifecycle    loaddata(data), revert()

A game is built from a ``levels.Level`` (the loader of the reference's ``.npz`` format lives there, as do
the per-level constants ``initial_colors`` / ``available_points`` / ``required_points``) and keeps that
level as its pristine copy, so ``revert()`` is a copy back, not a re-parse (files: ``levels.load_levels``).
Editing commands, saving, ``GameOfLife`` and ``AsyncGame`` are outside the hot path and not provided.
One env at a time, host arrays, every native call a (tiny) kernel launch: throughput lives in ``SafeLifeVectorEnv``.
"""
import os

import numpy as np

from . import levels as _levels
from . import speedups
from .cell_types import CellTypes
from .random import get_rng, set_rng

_AGENT_OR_EXIT = CellTypes.agent | CellTypes.exit


class SafeLifeGame(object):
    points_on_level_exit, game_over, file_name = 1, False, None

    def __init__(self, board_size=(10, 10), level=None):
        """``level``: a ``levels.Level``; without one, an empty board of ``board_size`` with the agent
        in its middle (what the reference's constructor yields)."""
        if level is None:
            h, w = board_size
            board = np.zeros((h, w), np.uint16)
            board[h // 2, w // 2] = CellTypes.player
            level = _levels.Level(board, agent_locs=[[h // 2, w // 2]])
        self._level = level
        self._seed = None
        self._rng = None
        self._install(level)

    # ------------------------------------------------------------------ construction / reset
    def _install(self, level):
        """(Re)start from the pristine level: state arrays are fresh copies, the per-level constants of
        the reward glue are derived once."""
        self.board = level.board.copy()
        self.goals = level.goals.copy()
        self.agent_locs = np.ascontiguousarray(level.agent_locs, dtype=np.int64).reshape(-1, 2)
        self.agent_names = np.array(["agent%d" % k for k in range(len(self.agent_locs))])
        self.spawn_prob = level.spawn_prob
        self.min_performance = level.min_performance
        self.points_table = level.points_table.copy()
        if len(self.points_table) != len(self.agent_locs):        # one table per agent
            self.points_table = np.resize(self.points_table, (len(self.agent_locs), 8, 9))
        flat = level.exit_locs
        self.exit_locs = np.unravel_index(flat, self.board.shape)
        self.num_steps, self.game_over = 0, False
        self._static_goals = None          # unknown until the goals have been advanced once
        self._counts = None                # cache of alive_counts, dropped whenever the board changes
        self._init_data = {"board": level.board}              # side_effect_score reads the starting board here
        self.initial_counts = self.alive_counts
        self.initial_colors = _levels.initial_colors(level.board)
        self.update_exit_colors()

    @classmethod
    def loaddata(cls, data, auto_cls=True):
        """From a dict / ``.npz`` record with the reference's keys.  The ``class`` key of level files is
        ignored: every level runs SafeLifeGame physics."""
        return cls(level=_levels.Level.from_data(data))

    def revert(self):
        self._install(self._level)
        return True                        # (the reference reports whether there was a state to go back to)

    # ------------------------------------------------------------------ identity, randomness
    @property
    def title(self):
        if not self.file_name:
            return None
        stem, _, ext = os.path.basename(self.file_name).rpartition(".")
        if not stem:
            return ext
        key = self._seed.spawn_key if self._seed is not None else ()
        return "%s-e%d" % (stem, key[-1]) if ext in ("yaml", "json") and key else stem

    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, value):
        self._seed = value if isinstance(value, np.random.SeedSequence) else np.random.SeedSequence(value)
        self._rng = np.random.default_rng(self._seed)          # PCG64: the only generator the kernels model

    @property
    def rng(self):
        return get_rng() if self._rng is None else self._rng

    # ------------------------------------------------------------------ agents
    def _agent_index(self):
        return self.agent_locs[:, 0], self.agent_locs[:, 1]

    def _agent_cells(self):
        return self.board[self._agent_index()]

    def agent_is_active(self):
        return (self._agent_cells() & CellTypes.agent) != 0

    def has_exited(self):
        """An agent that stepped into an exit leaves the exit cell behind at its location."""
        return (self._agent_cells() & _AGENT_OR_EXIT) == CellTypes.exit

    # ------------------------------------------------------------------ physics (the native path)
    def execute_actions(self, actions):
        if not (self.agent_locs.dtype == np.int64 and self.agent_locs.flags.c_contiguous):
            self.agent_locs = np.ascontiguousarray(self.agent_locs, dtype=np.int64)
        speedups.execute_actions(self.board, self.agent_locs, actions)
        self._counts = None

    def advance_board(self):
        """One CA step of the board under the game's own generator, then of the goals unless they are
        known not to change (decided after their first step: unchanged and without spawners)."""
        self.num_steps += 1
        self._counts = None
        with set_rng(self.rng):
            self.board = speedups.advance_board(self.board, self.spawn_prob)
            if self._static_goals:
                return
            stepped = speedups.advance_board(self.goals, self.spawn_prob)
            if self._static_goals is None:
                has_spawner = bool(np.any(stepped & CellTypes.spawning))
                self._static_goals = not has_spawner and bool(np.array_equal(stepped, self.goals))
            self.goals = stepped

    def update_exit_colors(self):
        """Agents that have earned the level's required points stand on an open exit; every exit cell of the
        level is repainted -- red once anybody may leave."""
        may_leave = self.can_exit()
        rows, cols = self._agent_index()
        cells = self.board[rows, cols] & np.uint16(~CellTypes.exit & 0xFFFF)
        self.board[rows, cols] = np.where(may_leave, cells | CellTypes.exit, cells)
        paint = CellTypes.level_exit | (CellTypes.color_r if np.any(may_leave) else 0)
        self.board[self.exit_locs] = paint

    # ------------------------------------------------------------------ scoring
    @property
    def alive_counts(self):
        """int64 [8 goal colours, 9 = 8 cell colours + empty] histogram (native), cached per board state."""
        if self._counts is None:
            counts = speedups.alive_counts(self.board, self.goals)
            counts.setflags(write=False)
            self._counts = counts
        return self._counts

    def _weighted(self, counts):
        """sum(points_table * counts) for every agent."""
        return np.einsum("agc,gc->a", self.points_table, counts)

    def _exit_bonus(self):
        return self.points_on_level_exit * self.has_exited()

    def current_points(self):
        return self._weighted(self.alive_counts) + self._exit_bonus()

    def points_earned(self):
        return self._weighted(self.alive_counts - self.initial_counts) + self._exit_bonus()

    def initial_available_points(self):
        return np.array([_levels.available_points(table, self.initial_counts, self.initial_colors)
                         for table in self.points_table], dtype=np.int64)

    def required_points(self):
        return np.array([_levels.required_points(self.min_performance, int(avail))
                         for avail in self.initial_available_points()], dtype=np.int64)

    def can_exit(self):
        enough = np.maximum(self.points_earned(), 0) >= self.required_points()
        return self.agent_is_active() & enough
