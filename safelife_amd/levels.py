"""
Level data: the on-disk ``.npz`` format of the reference and the *level pool* that feeds the
batched environment's on-device reset.

* ``load_levels(path)`` reads what ``SafeLifeGame.save`` / the benchmark archives contain
  (reference: safelife_game.py:200-234,615-637; archives with a structured ``levels`` array:
  level_iterator.py:89-99), including the legacy single-agent keys ``agent_loc`` (x, y) and
  ``orientation``.  The ``class`` key is ignored.
* ``LevelPool`` flattens a list of levels into the arrays ``sl_env_batch.pool_*`` expects and
  computes the per-level constants of the reward glue (safelife_game.py:665-714):
  ``initial_points``, ``required_points``.  The histogram those need comes from the device
  ``alive_counts`` kernel by default; host code here is only the integer glue around it.
"""
import os

import numpy as np

from .cell_types import CellTypes, DEFAULT_POINTS_TABLE

_MASK64 = (1 << 64) - 1


class Level(object):
    """One level instance = board + goals + agent(s) + constants + (optionally) its RNG seed."""

    def __init__(self, board, goals=None, agent_locs=None, spawn_prob=0.3, min_performance=-1.0,
                 points_table=None, orientation=None, name=None, seed=None, rng_words=None):
        self.board = np.array(board, dtype=np.uint16)
        if self.board.ndim != 2:
            raise ValueError("board must be 2-dimensional")
        self.goals = (np.zeros_like(self.board) if goals is None
                      else np.array(goals, dtype=np.uint16))
        if self.goals.shape != self.board.shape:
            raise ValueError("goals must have the board's shape")
        locs = np.empty((0, 2), np.int64) if agent_locs is None else np.array(agent_locs, np.int64)
        self.agent_locs = locs.reshape(-1, 2)
        self.spawn_prob = float(spawn_prob)
        self.min_performance = float(min_performance)
        if points_table is None:
            points_table = np.tile(DEFAULT_POINTS_TABLE, [len(self.agent_locs), 1, 1])
        self.points_table = np.array(points_table, dtype=np.int64).reshape(-1, 8, 9)
        if orientation is not None and len(self.agent_locs):
            # GameState.orientation setter (safelife_game.py:328-332)
            idx = tuple(self.agent_locs.T)
            bits = np.uint16((int(orientation) & 3) << CellTypes.orientation_bit)
            self.board[idx] = (self.board[idx] & ~CellTypes.orientation_mask) | bits
        self.name = name
        self.seed = seed
        self.rng_words = None if rng_words is None else np.array(rng_words, dtype=np.uint64)

    @property
    def shape(self):
        return self.board.shape

    @property
    def exit_locs(self):
        """Flat indices of GameState.exit_locs (safelife_game.py:533-535), row-major order."""
        mask = (self.board & (CellTypes.exit | CellTypes.agent)) == CellTypes.exit
        return np.flatnonzero(mask).astype(np.int32)

    def as_data(self):
        """dict in the reference's serialised form (safelife_game.py:200-209,615-620)."""
        return {"board": self.board.copy(), "goals": self.goals.copy(), "agent_locs": self.agent_locs.copy(),
                "spawn_prob": self.spawn_prob, "min_performance": self.min_performance,
                "points_table": self.points_table.copy()}

    def initial_rng_words(self):
        """PCG64 words of ``np.random.default_rng(seed)`` (safelife_game.py:175-180)."""
        if self.rng_words is not None:
            return self.rng_words
        seed = self.seed
        if not isinstance(seed, np.random.SeedSequence):
            seed = np.random.SeedSequence(seed)
        st = np.random.default_rng(seed).bit_generator.state["state"]
        s, i = st["state"], st["inc"]
        return np.array([s >> 64, s & _MASK64, i >> 64, i & _MASK64], dtype=np.uint64)

    @classmethod
    def from_data(cls, data, name=None, seed=None):
        """Build from a dict / NpzFile / structured-array record with the reference's keys."""
        names = getattr(getattr(data, "dtype", None), "names", None)
        keys = set(names) if names else set(data.keys())
        kw = {}
        if "agent_loc" in keys:                       # legacy: (x, y) of a single agent
            kw["agent_locs"] = np.array(data["agent_loc"], np.int64)[None, ::-1]
        elif "agent_locs" in keys:
            kw["agent_locs"] = data["agent_locs"]
        if "spawn_prob" in keys:
            kw["spawn_prob"] = float(data["spawn_prob"])
        if "min_performance" in keys:
            kw["min_performance"] = float(data["min_performance"])
        if "points_table" in keys:
            kw["points_table"] = data["points_table"]
        if "orientation" in keys:
            kw["orientation"] = int(data["orientation"])
        if name is None and "name" in keys:
            name = str(data["name"])
        goals = data["goals"] if "goals" in keys else None
        return cls(data["board"], goals, name=name, seed=seed, **kw)


def load_levels(path):
    """All levels of one ``.npz`` file (a single level or a benchmark archive)."""
    path = os.path.abspath(os.path.expanduser(path))
    base = path[:-4] if path.endswith(".npz") else path
    out = []
    with np.load(path, allow_pickle=False) as data:
        if "levels" in data.files:
            for rec in data["levels"]:
                lvl = Level.from_data(rec)
                lvl.name = os.path.join(base, lvl.name or str(len(out)))
                out.append(lvl)
        else:
            out.append(Level.from_data({k: data[k] for k in data.files}, name=path))
    return out


class SafeLifeLevelIterator(object):
    """Yields ``SafeLifeGame`` objects from ``.npz`` level files / archives, each with its own child
    of the iterator's SeedSequence (the static-file half of the reference's iterator,
    level_iterator.py:148-266).  Procedural generation (``.yaml`` specs) is outside the hot path: feed
    generated levels as files, or use ``LevelPool`` for the device-resident environments."""

    def __init__(self, *paths, repeat_levels=False, seed=None):
        self.levels = []
        for path in paths:
            for lv in load_levels(path):
                self.levels.append(lv)
        if not self.levels:
            raise FileNotFoundError("No levels found for %r" % (paths,))
        self.repeat_levels = repeat_levels
        self.idx = 0
        self.seed(seed)

    def seed(self, seed):
        if not isinstance(seed, np.random.SeedSequence):
            seed = np.random.SeedSequence(seed)
        self._seed = seed

    def __iter__(self):
        return self

    def __next__(self):
        from .game import SafeLifeGame
        if self.idx >= len(self.levels) and not self.repeat_levels:
            raise StopIteration
        lv = self.levels[self.idx % len(self.levels)]
        self.idx += 1
        game = SafeLifeGame.loaddata(lv.as_data())
        game.file_name = lv.name
        game.seed = self._seed.spawn(1)[0]
        return game


def _device_counts(boards, goals):
    """[L,8,9] int64 alive_counts of every level through the HIP kernel."""
    from . import speedups
    d_b = speedups._to_device(boards, np.uint16)
    d_g = speedups._to_device(goals, np.uint16)
    return speedups._to_host(speedups.alive_counts_batch(d_b, d_g), np.int64)


def initial_colors(board):
    """bool[9]: colours new cells can take (safelife_game.py:670-675); index 8 = 'empty'."""
    generators = CellTypes.agent | CellTypes.alive | CellTypes.spawning
    cols = (board[(board & generators) > 0] & CellTypes.rainbow_color) >> CellTypes.color_bit
    out = np.zeros(9, dtype=bool)
    out[np.unique(cols)] = True
    out[8] = True
    return out


def available_points(table, counts, colors):
    """GameWithGoals.initial_available_points for one agent (safelife_game.py:696-709)."""
    goal_counts = counts.sum(axis=1)
    best = (table * colors).max(axis=1)
    return int((best * goal_counts).sum() - (table * counts).sum())


def required_points(min_performance, available):
    """GameWithGoals.required_points (safelife_game.py:711-714): float64 ceil, clamped at 0."""
    return max(0, int(np.int64(np.ceil(np.float64(min_performance) * available))))


class LevelPool(object):
    """Flat arrays for ``sl_env_batch.pool_*`` (single-agent levels of one board shape)."""

    ARRAYS = ("pool_board", "pool_goals", "pool_agent_loc", "pool_exit_locs", "pool_rng",
              "pool_spawn_prob", "pool_required_reset", "pool_required_step",
              "pool_initial_points", "pool_table_idx", "points_table")

    def __init__(self, levels, *, min_performance_fraction=1.0, seed=None, counts_fn=None,
                 exit_slots=None, refreshable=False, n_agents=1):
        """
        levels : list of Level (same shape, at most one agent each)
        min_performance_fraction : the MinPerformanceScheduler factor (env_wrappers.py:142-145):
            steps after a reset require ``ceil(min_performance * fraction * available)`` points,
            while the reset itself (first observation) uses the level's own min_performance.
        seed : levels without an RNG of their own get children of this SeedSequence, in order
            (as SafeLifeLevelIterator.fill_queue does, level_iterator.py:218).
        counts_fn : (boards[L,H,W], goals[L,H,W]) -> int64 [L,8,9]; defaults to the HIP kernel.
        n_agents : > 1 -- a pool of MULTI-agent levels (exactly that many agents each: the reference's
            levels/random/multi-agent specs) for ``multi_env.SafeLifeMultiAgentVectorEnv``: the per-agent
            constants (locations, points tables, initial and required points: safelife_game.py:684-714 per agent)
            are kept in ``pool_agent_*`` arrays [L, A, ...]; the single-agent arrays describe agent 0.
        refreshable : the pool can take NEW levels while envs are stepping (``replace`` here,
            ``SafeLifeVectorEnv.pool_stage`` / ``pool_commit`` on the device) -- the device-resident
            counterpart of the reference's level iterator handing every reset a fresh level
            (level_iterator.py:200-223, safelife_env.py:203-218).  Every logical level l then has TWO
            physical slots, l and l + L: a replacement is written into the one that is not current, and
            the successor table (``next_table``) is switched to it between two steps, so a reset that
            is loading the old content never sees a half-written slot and envs that are still playing
            it keep an intact copy until the SAME level is replaced once more.
        """
        if not levels:
            raise ValueError("empty level list")
        shape = levels[0].shape
        L = len(levels)
        for lv in levels:
            if lv.shape != shape:
                raise ValueError("all levels of a pool must share one board shape")
            if int(n_agents) > 1:
                if len(lv.agent_locs) != int(n_agents) or len(lv.points_table) != int(n_agents):
                    raise ValueError("a multi-agent pool holds levels with exactly n_agents agents (and one points table each)")
            elif len(lv.agent_locs) > 1:
                raise ValueError("a single-agent pool (LevelPool(n_agents=A) + SafeLifeMultiAgentVectorEnv for more)")
        self.n_agents = int(n_agents)
        if self.n_agents > 1 and refreshable:
            raise ValueError("refreshable pools are single-agent")
        self.levels = list(levels)
        self.shape = shape
        H, W = shape
        exits = [lv.exit_locs for lv in levels]
        E = max(1, max(len(x) for x in exits)) if exit_slots is None else int(exit_slots)
        if any(len(x) > E for x in exits):
            raise ValueError("exit_slots too small")

        self.pool_board = np.stack([lv.board for lv in levels])
        self.pool_goals = np.stack([lv.goals for lv in levels])
        #: spawners are never created by the rules, so a pool without any stays free of random draws
        self.has_spawner = bool(((self.pool_board | self.pool_goals) & CellTypes.spawning).any())
        self.pool_agent_loc = np.full((L, 2), -1, np.int32)
        self.pool_exit_locs = np.full((L, E), -1, np.int32)
        self.pool_spawn_prob = np.array([lv.spawn_prob for lv in levels], np.float32)
        seq = seed if isinstance(seed, np.random.SeedSequence) else np.random.SeedSequence(seed)
        rng = []
        for k, lv in enumerate(levels):
            if len(lv.agent_locs):
                self.pool_agent_loc[k] = lv.agent_locs[0]
            self.pool_exit_locs[k, :len(exits[k])] = exits[k]
            if lv.rng_words is None and lv.seed is None:
                lv = Level(lv.board, lv.goals, lv.agent_locs, lv.spawn_prob, lv.min_performance,
                           lv.points_table, seed=seq.spawn(1)[0])
            rng.append(lv.initial_rng_words())
        self.pool_rng = np.stack(rng).astype(np.uint64)

        # de-duplicated points tables (every agent's, agent 0's first)
        tables, idx = [], []

        def table_index(t):
            for j, u in enumerate(tables):
                if np.array_equal(t, u):
                    return j
            tables.append(t)
            return len(tables) - 1
        for lv in levels:
            idx.append(table_index((lv.points_table[0] if len(lv.points_table) else DEFAULT_POINTS_TABLE).astype(np.int32)))
        agent_idx = [[table_index(lv.points_table[a].astype(np.int32)) for a in range(self.n_agents)] for lv in levels] \
            if self.n_agents > 1 else None
        self.points_table = np.stack(tables).astype(np.int32)
        self.pool_table_idx = np.array(idx, np.int32)

        counts = (counts_fn or _device_counts)(self.pool_board, self.pool_goals)
        counts = np.asarray(counts, np.int64).reshape(L, 8, 9)
        self.initial_counts = counts
        self.pool_initial_points = np.zeros(L, np.int32)
        self.pool_required_reset = np.zeros(L, np.int32)
        self.pool_required_step = np.zeros(L, np.int32)
        frac = float(min_performance_fraction)
        for k, lv in enumerate(levels):
            table = self.points_table[idx[k]].astype(np.int64)
            self.pool_initial_points[k] = int((table * counts[k]).sum())
            avail = available_points(table, counts[k], initial_colors(lv.board))
            self.pool_required_reset[k] = required_points(lv.min_performance, avail)
            # game.min_performance *= fraction  (float64 product, then the same ceil)
            self.pool_required_step[k] = required_points(np.float64(lv.min_performance) * frac, avail)

        if self.n_agents > 1:       # the same per agent: its own table against the level's counts
            A = self.n_agents
            self.pool_agent_locs = np.stack([lv.agent_locs for lv in levels]).astype(np.int32).reshape(L, A, 2)
            self.pool_agent_table_idx = np.array(agent_idx, np.int32).reshape(L, A)
            self.pool_agent_initial_points = np.zeros((L, A), np.int32)
            self.pool_agent_required_reset = np.zeros((L, A), np.int32)
            self.pool_agent_required_step = np.zeros((L, A), np.int32)
            for k, lv in enumerate(levels):
                colors = initial_colors(lv.board)
                for a in range(A):
                    table = self.points_table[agent_idx[k][a]].astype(np.int64)
                    self.pool_agent_initial_points[k, a] = int((table * counts[k]).sum())
                    avail = available_points(table, counts[k], colors)
                    self.pool_agent_required_reset[k, a] = required_points(lv.min_performance, avail)
                    self.pool_agent_required_step[k, a] = required_points(np.float64(lv.min_performance) * frac, avail)

        self._frac, self._seq, self._counts_fn = frac, seq, counts_fn
        self.refreshable = bool(refreshable)
        self.bank = np.zeros(L, np.int8)              # refreshable: which of its two slots holds level l now
        self.slot_version = np.zeros(2 * L, np.int64) # how often each PHYSICAL slot has been rewritten (replace())
        if self.refreshable:                          # slots L .. 2L-1: the spare bank (starts as a copy: valid content)
            for k in self.ARRAYS:
                if k != "points_table":
                    setattr(self, k, np.concatenate([getattr(self, k)] * 2, axis=0))
            self.initial_counts = np.concatenate([self.initial_counts] * 2, axis=0)

    def __len__(self):
        """Levels of the pool (logical: a refreshable pool has twice as many physical slots)."""
        return len(self.levels)

    @property
    def n_slots(self):
        """Physical slots = the leading dimension of the ``pool_*`` arrays (``sl_env_batch.L``)."""
        return self.pool_board.shape[0]

    @property
    def exit_slots(self):
        return self.pool_exit_locs.shape[1]

    def arrays(self):
        return {k: getattr(self, k) for k in self.ARRAYS}

    def slot_of(self, level):
        """Physical slot that holds logical level `level` now."""
        level = int(level)
        return level + len(self) * int(self.bank[level])

    def next_table(self, level_stride=1):
        """int32 [n_slots] for ``sl_env_batch.pool_next``: an env on slot s (either bank of level s % L) loads the
        CURRENT slot of level (s % L + level_stride) % L at its next reset."""
        L = len(self)
        cur = np.arange(L, dtype=np.int64) + L * self.bank.astype(np.int64)
        succ = cur[(np.arange(self.n_slots) % L + int(level_stride)) % L]
        return succ.astype(np.int32)

    def prepare(self, levels):
        """Everything ``replace`` needs of `levels`, computed now: the checks, the cell counts (the HIP kernel, one batch),
        the points and the RNG words.  Level generation happens away from the stepping thread (the reference's level
        iterator fills its queue ahead, level_iterator.py:200-223); this belongs with it -- ``replace`` / ``pool_stage``
        of a PreparedLevels only copy.  ``take(idx)`` picks a subset."""
        if not self.refreshable:
            raise ValueError("build the pool with refreshable=True")
        levels = list(levels)
        n, E = len(levels), self.exit_slots
        for lv in levels:
            if lv.shape != self.shape or len(lv.agent_locs) > 1 or len(lv.exit_locs) > E:
                raise ValueError("the new level does not fit the pool (shape, agents or exit slots)")
            if not self.has_spawner and bool(((lv.board | lv.goals) & CellTypes.spawning).any()):
                raise ValueError("a level with spawners cannot join a pool that was built spawner-free")
        pl = PreparedLevels()
        pl.pool_board = np.stack([lv.board for lv in levels])
        pl.pool_goals = np.stack([lv.goals for lv in levels])
        counts = np.asarray((self._counts_fn or _device_counts)(pl.pool_board, pl.pool_goals), np.int64).reshape(n, 8, 9)
        pl.initial_counts = counts
        pl.pool_agent_loc = np.full((n, 2), -1, np.int32)
        pl.pool_exit_locs = np.full((n, E), -1, np.int32)
        pl.pool_spawn_prob = np.array([lv.spawn_prob for lv in levels], np.float32)
        pl.pool_rng = np.zeros((n, 4), np.uint64)
        pl.pool_table_idx = np.zeros(n, np.int32)
        pl.pool_initial_points = np.zeros(n, np.int32)
        pl.pool_required_reset = np.zeros(n, np.int32)
        pl.pool_required_step = np.zeros(n, np.int32)
        pl.levels = []
        for i, lv in enumerate(levels):
            t = (lv.points_table[0] if len(lv.points_table) else DEFAULT_POINTS_TABLE).astype(np.int32)
            idx = [j for j, u in enumerate(self.points_table) if np.array_equal(t, u)]
            if not idx:
                raise ValueError("the new level's points table is not one of the pool's (the score tables are built once)")
            if lv.rng_words is None and lv.seed is None:
                lv = Level(lv.board, lv.goals, lv.agent_locs, lv.spawn_prob, lv.min_performance,
                           lv.points_table, seed=self._seq.spawn(1)[0])
            if len(lv.agent_locs):
                pl.pool_agent_loc[i] = lv.agent_locs[0]
            ex = lv.exit_locs
            pl.pool_exit_locs[i, :len(ex)] = ex
            pl.pool_rng[i] = lv.initial_rng_words()
            pl.pool_table_idx[i] = idx[0]
            table = self.points_table[idx[0]].astype(np.int64)
            pl.pool_initial_points[i] = int((table * counts[i]).sum())
            avail = available_points(table, counts[i], initial_colors(lv.board))
            pl.pool_required_reset[i] = required_points(lv.min_performance, avail)
            pl.pool_required_step[i] = required_points(np.float64(lv.min_performance) * self._frac, avail)
            pl.levels.append(lv)
        return pl

    def replace(self, slots, levels):
        """New content for the logical levels `slots` (refreshable pools): written into each level's SPARE slot, which
        becomes its current one.  Returns the physical slots written.  Host arrays only -- the device copy and the
        switch of the successor table are ``SafeLifeVectorEnv.pool_stage`` / ``pool_commit``.  `levels`: a list of
        Level, or what ``prepare`` made of one (then this only copies).  A new level must fit the pool as built: same
        shape, no more exits than ``exit_slots``, a points table the pool already has, and no spawner in a pool that was
        built without any (the kernels were chosen for that)."""
        if not self.refreshable:
            raise ValueError("build the pool with refreshable=True")
        slots = np.asarray([int(x) for x in slots], np.int64)
        pl = levels if isinstance(levels, PreparedLevels) else self.prepare(levels)
        L = len(self)
        if len(slots) != len(pl.levels) or len(set(slots.tolist())) != len(slots):
            raise ValueError("one new level per distinct slot")
        if len(slots) and (slots.min() < 0 or slots.max() >= L):
            raise ValueError("no such level slot: %d" % (slots.min() if slots.min() < 0 else slots.max()))
        phys = slots + L * (1 - self.bank[slots].astype(np.int64))
        for k in PreparedLevels.ARRAYS:
            getattr(self, k)[phys] = getattr(pl, k)
        self._replaced = (slots.copy(), [self.levels[l] for l in slots.tolist()])      # (for undo_replace)
        for l, lv in zip(slots.tolist(), pl.levels):
            self.levels[l] = lv
        self.bank[slots] = 1 - self.bank[slots]
        self.slot_version[phys] += 1
        return phys.tolist()

    def undo_replace(self):
        """The last ``replace()`` did not reach the device (its copy failed): the levels' previous slots are current
        again.  (The spare slots keep the new rows -- nothing refers to them.)"""
        slots, old = self._replaced
        self.bank[slots] = 1 - self.bank[slots]
        for l, lv in zip(slots.tolist(), old):
            self.levels[l] = lv
        self._replaced = (np.zeros(0, np.int64), [])


class PreparedLevels(object):
    """``LevelPool.prepare``'s result: the rows the pool's arrays take for some new levels."""

    ARRAYS = ("pool_board", "pool_goals", "pool_agent_loc", "pool_exit_locs", "pool_rng", "pool_spawn_prob",
              "pool_required_reset", "pool_required_step", "pool_initial_points", "pool_table_idx", "initial_counts")

    def __len__(self):
        return len(self.levels)

    def take(self, idx):
        """The levels `idx` of this set (repeats allowed), as a PreparedLevels."""
        idx = np.asarray(idx, np.int64)
        out = PreparedLevels()
        for k in self.ARRAYS:
            setattr(out, k, getattr(self, k)[idx])
        out.levels = [self.levels[int(i)] for i in idx]
        return out


def empty_env_arrays(pool, num_envs):
    """Zero-initialised host mirrors of the per-env arrays of ``sl_env_batch`` (names as in
    include/safelife_hip.h).  Used by tests to drive the oracle and the device from one layout."""
    H, W = pool.shape
    B, E = int(num_envs), pool.exit_slots
    a = {
        "board": np.zeros((B, H, W), np.uint16), "goals": np.zeros((B, H, W), np.uint16),
        "agent_loc": np.full((B, 2), -1, np.int32), "exit_locs": np.full((B, E), -1, np.int32),
        "rng": np.zeros((B, 4), np.uint64), "spawn_prob": np.zeros(B, np.float32),
        "num_steps": np.zeros(B, np.int32), "old_value": np.zeros(B, np.int32),
        "required_points": np.zeros(B, np.int32), "initial_points": np.zeros(B, np.int32),
        "table_idx": np.zeros(B, np.int32), "goals_static": np.zeros(B, np.uint8),
        "is_active": np.zeros(B, np.uint8), "episode_reward": np.zeros(B, np.float32),
        "episode_length": np.zeros(B, np.int32), "level_idx": np.zeros(B, np.int32),
        "episode_idx": np.zeros(B, np.int32), "loaded": np.zeros(B, np.uint8),
        "reward": np.zeros(B, np.float32), "done": np.zeros(B, np.uint8),
        "success": np.zeros(B, np.uint8), "times_up": np.zeros(B, np.uint8),
    }
    a.update({k: np.ascontiguousarray(v) for k, v in pool.arrays().items()})
    return a
