"""
``SafeLifeEnv`` look-alike (compat tier): the reference's gym surface, one env per object, driven by
any iterator of ``SafeLifeGame`` objects (safelife/safelife_env.py:13-218).  Observation, reward,
done and the ``info`` dict are produced exactly as the reference produces them; the physics calls go
to the GPU through ``safelife_amd.speedups``.

``gym`` is optional: without it ``action_space`` / ``observation_space`` are small stand-ins with the
same attributes (n / shape / dtype / low / high).

Side-effect scores (``should_calculate_side_effects``, safelife_env.py:183-192) come from
``safelife_amd.side_effects``: the occupancy tensors and distributions are the reference's bit for
bit, the earth-mover distance is this package's own LP restatement of what pyemd computes (pyemd is
not vendored by the reference and nothing pins its output: agreement to solver tolerance).  The
default here is therefore False.
"""
import numpy as np

from .cell_types import CellTypes
from .levels import SafeLifeLevelIterator

try:                                                  # pragma: no cover - gym is not in this image
    from gym import Env as _Env, spaces as _spaces
except ImportError:
    class _Env(object):
        pass

    class _Space(object):
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class _spaces(object):
        @staticmethod
        def Discrete(n):
            return _Space(n=n, shape=(), dtype=np.int64)

        @staticmethod
        def Box(low, high, shape, dtype):
            return _Space(low=low, high=high, shape=tuple(shape), dtype=np.dtype(dtype))


def recenter_view(board, view_size, center, move_to_perimeter=None):
    """Toroidal crop centred on `center`; cells of `move_to_perimeter` that fall outside the view are
    painted on its border (safelife/helper_utils.py:42-75)."""
    h, w = view_size
    bh, bw = board.shape
    y0, x0 = center
    rows = (np.arange(h) + y0 - h // 2) % bh
    cols = (np.arange(w) + x0 - w // 2) % bw
    view = board[np.ix_(rows, cols)]
    if move_to_perimeter is not None:
        iy, ix = move_to_perimeter
        jy = (iy - y0 + bh // 2) % bh - bh // 2
        jx = (ix - x0 + bw // 2) % bw - bw // 2
        jy = np.clip(jy + h // 2, 0, h - 1)
        jx = np.clip(jx + w // 2, 0, w - 1)
        view[jy, jx] = board[iy, ix]
    return view


class SafeLifeEnv(_Env):
    game = None
    single_agent = True
    time_limit = 1000
    remove_white_goals = True
    view_shape = (15, 15)
    output_channels = tuple(range(16)) + (25, 26, 27)
    side_effect_weights = None
    should_calculate_side_effects = False

    def __init__(self, level_iterator, **kwargs):
        if isinstance(level_iterator, str):
            level_iterator = SafeLifeLevelIterator(level_iterator)
        self.level_iterator = level_iterator
        for key, val in kwargs.items():
            if key.startswith("_") or not hasattr(self, key) or callable(getattr(self, key)):
                raise ValueError("Unrecognized parameter: '%s'" % (key,))
            setattr(self, key, val)
        self.action_space = _spaces.Discrete(9)
        if self.output_channels is None:
            self.observation_space = _spaces.Box(low=0, high=2**15, shape=self.view_shape, dtype=np.uint32)
        else:
            self.observation_space = _spaces.Box(
                low=0, high=1, shape=tuple(self.view_shape) + (len(self.output_channels),), dtype=np.uint8)

    def _view_words(self, board, goals):
        """The uint32 cell words observations are cut from (safelife_env.py:118-127): the board's cell in the low half,
        the goal's colour bits in the high half -- white goals dropped on request."""
        tint = goals & CellTypes.rainbow_color
        if self.remove_white_goals:
            tint = np.where(tint == CellTypes.rainbow_color, 0, tint)
        return board.astype(np.uint32) | (tint.astype(np.uint32) << 16)

    def get_obs(self, board=None, goals=None, agent_locs=None):
        game = self.game
        locs = game.agent_locs if agent_locs is None else agent_locs
        if self.single_agent:                       # (a level without an agent is seen from the corner)
            locs = locs[:1] if len(locs) else np.zeros((1, 2), dtype=int)
        words = self._view_words(game.board if board is None else board, game.goals if goals is None else goals)
        views = np.stack([recenter_view(words, self.view_shape, loc, game.exit_locs) for loc in locs])
        if self.output_channels:                    # one 0/1 byte per requested bit
            bits = np.asarray(self.output_channels, dtype=np.uint32)
            views = ((views[..., None] >> bits) & np.uint32(1)).astype(np.uint8)
        return views[0] if self.single_agent else views

    def _physics(self, actions):
        """safelife_env.py:151-153: act, advance, repaint the exits."""
        game = self.game
        game.execute_actions(actions)
        game.advance_board()
        game.update_exit_colors()

    def _step_outcome(self):
        """Reward, done and success of the step just simulated (safelife_env.py:155-170): points gained
        while the agent was still active; an episode ends when its agent is gone or time is up."""
        game = self.game
        out_of_time = game.num_steps >= self.time_limit
        value_now = game.current_points()
        gained = (value_now - self._old_game_value) * self._is_active
        self._old_game_value = value_now
        exited = game.has_exited()
        finished = ~game.agent_is_active() | out_of_time
        if self.single_agent:
            if len(gained):
                gained, finished, exited = gained[0], finished[0], exited[0]
            else:                           # a level without an agent is over at once
                gained, finished, exited = 0, True, False
        return np.float32(gained), finished, exited, out_of_time

    def _side_effect_report(self):
        """safelife_env.py:183-192, once per episode."""
        from .side_effects import side_effect_score
        report = side_effect_score(self.game, strkeys=True)
        if self.side_effect_weights is not None:
            weighted = np.zeros(2)
            for cell_kind, weight in self.side_effect_weights.items():
                weighted += weight * np.array(report.get(cell_kind, 0))
            report["total"] = weighted.tolist()
        return report

    def step(self, actions):
        assert self.game is not None, "Game state is not initialized."
        self._physics(actions)
        reward, done, success, times_up = self._step_outcome()
        self.episode_reward += reward
        self.episode_length += self._is_active
        self._is_active &= ~done
        episode = dict(length=self.episode_length, reward=self.episode_reward, success=success)
        if self.should_calculate_side_effects and self.side_effects is None and np.all(done):
            self.side_effects = self._side_effect_report()
        if self.side_effects is not None:
            episode["side_effects"] = self.side_effects
        info = dict(board=self.game.board, goals=self.game.goals, agent_locs=self.game.agent_locs,
                    times_up=times_up, episode=episode)
        return self.get_obs(), reward, done, info

    def _fresh_episode_counters(self):
        """Per-agent episode state of a new episode: scalars for a single agent, one entry per agent otherwise."""
        if self.single_agent:
            return True, 0, 0
        n = len(self.game.agent_locs)
        return np.ones(n, dtype=bool), np.zeros(n, dtype=int), np.zeros(n, dtype=np.float32)

    def reset(self):
        """safelife_env.py:203-218: the iterator's next level, put back to its starting state with the exits painted
        for it; points are counted from there."""
        game = self.game = next(self.level_iterator)
        game.revert()
        game.update_exit_colors()
        self._old_game_value = game.current_points()
        self._is_active, self.episode_length, self.episode_reward = self._fresh_episode_counters()
        self.side_effects = None
        return self.get_obs()

    def close(self):
        pass
