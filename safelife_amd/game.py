"""
``SafeLifeGame`` look-alike on top of the GPU ``speedups`` functions (compat tier).

Same attribute names, dtypes and call sequence as the reference's
``GameState -> GameWithGoals -> SafeLifeGame`` chain (safelife/safelife_game.py:126-761), restricted
to what ``SafeLifeEnv.step()/reset()``, the reward wrappers and the episode logger touch:

    board, goals, agent_locs, agent_names, exit_locs, spawn_prob, min_performance, num_steps,
    points_table, points_on_level_exit, file_name, title, seed, rng, game_over,
    execute_actions(), advance_board(), update_exit_colors(), alive_counts, current_points(),
    points_earned(), initial_available_points(), required_points(), can_exit(), has_exited(),
    agent_is_active(), is_stochastic, revert(), serialize(), deserialize(), loaddata(), load(), save()

One env at a time, host arrays, every native call a (tiny) kernel launch: this tier exists so that
reference-style drivers and wrappers run unchanged; throughput lives in ``SafeLifeVectorEnv``.
Editing (`execute_edit`), `GameOfLife` and `AsyncGame` are outside the hot path and not provided.
"""
import os

import numpy as np

from . import speedups
from .cell_types import CellTypes, DEFAULT_POINTS_TABLE
from .random import get_rng, set_rng


class SafeLifeGame(object):
    spawn_prob = 0.3
    board = None
    goals = None
    file_name = None
    game_over = False
    points_on_level_exit = +1
    num_steps = 0
    min_performance = -1
    _seed = None
    _rng = None
    _static_goals = None
    default_points_table = DEFAULT_POINTS_TABLE

    def __init__(self, board_size=(10, 10)):
        self.exit_locs = (np.array([], dtype=int), np.array([], dtype=int))
        self.agent_locs = np.empty((0, 2), dtype=int)
        self.agent_names = np.array([], dtype=str)
        if board_size is not None:
            self.board = np.zeros(board_size, dtype=np.uint16)
            self.agent_locs = np.array(board_size).reshape(1, 2) // 2
            self.agent_names = np.array(["agent0"])
            self.board[self.agent_locs_idx] = CellTypes.player
            self.goals = np.zeros_like(self.board)
            self._needs_new_counts = True
            self.reset_points_table()
            self.setup_initial_counts()
            self._init_data = self.serialize()

    # ------------------------------------------------------------------ seeding (safelife_game.py:166-192)
    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, seed):
        if not isinstance(seed, np.random.SeedSequence):
            seed = np.random.SeedSequence(seed)
        self._seed = seed
        self._rng = np.random.default_rng(seed)

    @property
    def rng(self):
        return self._rng if self._rng is not None else get_rng()

    # ------------------------------------------------------------------ (de)serialisation
    def serialize(self):
        cls = self.__class__
        return {
            "spawn_prob": self.spawn_prob,
            "agent_locs": self.agent_locs.copy(),
            "agent_names": self.agent_names.copy(),
            "board": self.board.copy(),
            "class": "%s.%s" % (cls.__module__, cls.__name__),
            "goals": self.goals.copy(),
            "points_table": self.points_table.copy(),
            "min_performance": self.min_performance,
        }

    def deserialize(self, data, as_initial_state=True):
        keys = data.dtype.fields if getattr(getattr(data, "dtype", None), "fields", None) else data
        if as_initial_state:
            self._init_data = data
        self.board = np.array(data["board"], dtype=np.uint16)
        if "spawn_prob" in keys:
            self.spawn_prob = float(data["spawn_prob"])
        if "agent_loc" in keys:             # legacy single-agent key, stored as (x, y)
            self.agent_locs = np.ascontiguousarray(np.array(data["agent_loc"])[None, ::-1])
        elif "agent_locs" in keys:
            self.agent_locs = np.array(data["agent_locs"]).reshape(-1, 2)
        if "agent_names" in keys:
            self.agent_names = np.array(data["agent_names"])
        else:
            self.agent_names = np.array(["agent%i" % i for i in range(len(self.agent_locs))])
        if "orientation" in keys:
            self.orientation = int(data["orientation"])
        self.update_exit_locs()
        self.game_over = False
        self.num_steps = 0
        self.goals = np.array(data["goals"], dtype=np.uint16) if "goals" in keys else np.zeros_like(self.board)
        if "min_performance" in keys:
            self.min_performance = data["min_performance"]
        if "points_table" in keys:
            self.points_table = np.array(data["points_table"])
        else:
            self.reset_points_table()
        self._needs_new_counts = True
        if as_initial_state:
            self.setup_initial_counts()
        self._static_goals = None
        self.update_exit_colors()

    def revert(self):
        if hasattr(self, "_init_data"):
            self.deserialize(self._init_data)
            return True
        return False

    @classmethod
    def loaddata(cls, data, auto_cls=True):
        """The `class` key of level files is ignored: every level runs SafeLifeGame physics."""
        obj = cls(board_size=None)
        obj.deserialize(data)
        return obj

    @classmethod
    def load(cls, file_name, auto_cls=True):
        file_name = os.path.abspath(os.path.expanduser(file_name))
        with np.load(file_name) as data:
            obj = cls.loaddata({k: data[k] for k in data.files})
        obj.file_name = file_name
        return obj

    def save(self, file_name=None):
        file_name = file_name or self.file_name
        if file_name is None:
            raise ValueError("Must specify a file name")
        file_name = os.path.abspath(os.path.expanduser(file_name))
        if not file_name.endswith(".npz"):
            file_name += ".npz"
        self.file_name = file_name
        self._init_data = self.serialize()
        self.num_steps = 0
        np.savez_compressed(file_name, **self._init_data)

    # ------------------------------------------------------------------ simple properties
    @property
    def width(self):
        return self.board.shape[1]

    @property
    def height(self):
        return self.board.shape[0]

    @property
    def title(self):
        if self.file_name is None:
            return None
        fname = os.path.split(self.file_name)[-1]
        fname, *ext = fname.rsplit(".", 1)
        if ext and ext[0] in ("json", "yaml") and self._seed and self._seed.spawn_key:
            fname += "-e" + str(self._seed.spawn_key[-1])
        return fname

    @property
    def agent_locs_idx(self):
        return tuple(self.agent_locs.T)

    @property
    def orientation(self):
        agents = self.board[self.agent_locs_idx]
        return ((agents & CellTypes.orientation_mask) >> CellTypes.orientation_bit).astype(np.int64)

    @orientation.setter
    def orientation(self, value):
        value = (np.array(value, dtype=np.uint16) & 3) << CellTypes.orientation_bit
        idx = self.agent_locs_idx
        self.board[idx] = (self.board[idx] & ~CellTypes.orientation_mask) | value

    @property
    def is_stochastic(self):
        return bool((self.board & CellTypes.spawning).any())

    # ------------------------------------------------------------------ actions / physics
    def execute_actions(self, actions):
        """safelife_game.py:380-389 -> C execute_actions (advance_board.c:217-300)."""
        if self.agent_locs.dtype != np.int64 or not self.agent_locs.flags.c_contiguous:
            self.agent_locs = np.ascontiguousarray(self.agent_locs, dtype=np.int64)
        speedups.execute_actions(self.board, self.agent_locs, actions)

    def advance_board(self):
        """safelife_game.py:746-761: one CA step of the board, and of the goals unless static,
        drawing from the game's own generator."""
        with set_rng(self.rng):
            self.num_steps += 1
            self._needs_new_counts = True
            self.board = speedups.advance_board(self.board, self.spawn_prob)
            if not self._static_goals:
                new_goals = speedups.advance_board(self.goals, self.spawn_prob)
                if self._static_goals is None:
                    self._static_goals = bool(
                        not (new_goals & CellTypes.spawning).any() and (new_goals == self.goals).all())
                self.goals = new_goals

    # ------------------------------------------------------------------ exits (safelife_game.py:505-552)
    def has_exited(self):
        agents = self.board[self.agent_locs_idx]
        return agents & (CellTypes.agent | CellTypes.exit) == CellTypes.exit

    def agent_is_active(self):
        return self.board[self.agent_locs_idx] & CellTypes.agent > 0

    def update_exit_locs(self):
        exits = self.board & (CellTypes.exit | CellTypes.agent) == CellTypes.exit
        self.exit_locs = np.nonzero(exits)

    def update_exit_colors(self):
        can_exit = self.can_exit()
        idx = self.agent_locs_idx
        self.board[idx] = (self.board[idx] & ~CellTypes.exit) | (CellTypes.exit * can_exit).astype(np.uint16)
        exit_type = CellTypes.level_exit | CellTypes.color_r if can_exit.any() else CellTypes.level_exit
        self.board[self.exit_locs] = exit_type

    # ------------------------------------------------------------------ points (safelife_game.py:657-719)
    @property
    def alive_counts(self):
        if getattr(self, "_needs_new_counts", True):
            self._needs_new_counts = False
            self._alive_counts = speedups.alive_counts(self.board, self.goals)
            self._alive_counts.setflags(write=False)
        return self._alive_counts

    def setup_initial_counts(self):
        self.initial_counts = self.alive_counts
        self.initial_colors = np.zeros(9, dtype=bool)
        generators = CellTypes.agent | CellTypes.alive | CellTypes.spawning
        colors = self.board[self.board & generators > 0] & CellTypes.rainbow_color
        self.initial_colors[np.unique(colors) >> CellTypes.color_bit] = True
        self.initial_colors[-1] = True

    def reset_points_table(self):
        self.points_table = np.tile(self.default_points_table, [len(self.agent_locs), 1, 1])

    def _exit_points(self):
        return self.points_on_level_exit * self.has_exited()

    def current_points(self):
        points = (self.points_table * self.alive_counts).reshape(-1, 72)
        return np.sum(points, axis=1) + self._exit_points()

    def points_earned(self):
        delta = self.alive_counts - self.initial_counts
        points = (self.points_table * delta).reshape(-1, 72)
        return np.sum(points, axis=1) + self._exit_points()

    def initial_available_points(self):
        goal_counts = np.sum(self.initial_counts, axis=1)
        max_points = np.max(self.points_table * self.initial_colors, axis=2)
        total_available = np.sum(max_points * goal_counts, axis=1)
        initial_points = np.sum((self.points_table * self.initial_counts).reshape(-1, 72), axis=1)
        return total_available - initial_points

    def required_points(self):
        req_points = self.min_performance * self.initial_available_points()
        return np.maximum(0, np.int64(np.ceil(req_points)))

    def can_exit(self):
        points_earned = np.maximum(0, self.points_earned())
        is_agent = self.board[self.agent_locs_idx] & CellTypes.agent > 0
        return is_agent & (points_earned >= self.required_points())
