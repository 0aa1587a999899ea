"""
SafeLifeMultiAgentVectorEnv -- B device-resident envs with A agents each: the batched counterpart of the reference's
``SafeLifeEnv(single_agent=False)`` (safelife_env.py:148-218 with the unwrapping of :162-170 not taken).

Every agent has its own points table, exit condition, reward, done flag, episode accumulators and an observation centred
on itself; the agents' actions are applied in index order (advance_board.c:217-220) and the exits turn red when ANY of
them may leave (safelife_game.py:537-552).  One fused launch per step (``slhip_env_step_multi``: the size-generic kernel
family, one workgroup per board, any board shape); an env reloads its next pool level inside the step once ALL its
agents are done (training/base_algo.py:231-236).  Levels: the reference's ``levels/random/multi-agent`` specs, or any
level with A agents -- ``LevelPool(levels, n_agents=A)``.
"""
import ctypes as C

import numpy as np

from . import _hip
from .levels import LevelPool
from .vector_env import SafeLifeVectorEnv


class SafeLifeMultiAgentVectorEnv(object):
    """
    pool : LevelPool(levels, n_agents=A)
    num_envs, time_limit, remove_white_goals, view_shape, output_channels, auto_reset, first_level, level_stride,
    env_offset, episode_streams, points_on_level_exit : as SafeLifeVectorEnv.

    ``reset()`` -> obs;  ``step(actions)`` -> (obs, reward, done, info) with
        actions  int   [B, A]   0..8 (an agent that is done takes 0: training/base_algo.py:216-219)
        obs      uint8 [B, A, view_h, view_w, C]  (int32 [B, A, view_h, view_w] holding the uint32 view for
                 ``output_channels=None``)
        reward   float32 [B, A];  done uint8 [B, A]
        info     success, times_up uint8 [B, A]; episode_reward float32, episode_length int32 [B, A] (of the episode the
                 step belonged to, before any reload)
    all views of device tensors that the next step overwrites.
    """

    def __init__(self, pool, num_envs, *, time_limit=1000, remove_white_goals=True, view_shape=(15, 15),
                 output_channels=tuple(range(16)) + (25, 26, 27), auto_reset=True, first_level=None, level_stride=1,
                 env_offset=0, with_obs=True, points_on_level_exit=1, episode_streams=True):
        import torch
        if not isinstance(pool, LevelPool) or getattr(pool, "n_agents", 1) < 1:
            raise TypeError("pool must be a LevelPool")
        A = int(pool.n_agents)
        if A > _hip.SL_MAX_AGENTS:
            raise ValueError("at most %d agents per board" % _hip.SL_MAX_AGENTS)
        # the boards, goals, generators, exit tables, the pool and the view are the single-agent env's: built once there
        self.base = base = SafeLifeVectorEnv(pool, num_envs, time_limit=time_limit, remove_white_goals=remove_white_goals,
                                             view_shape=view_shape, output_channels=output_channels, auto_reset=auto_reset,
                                             first_level=first_level, level_stride=level_stride, env_offset=env_offset,
                                             with_obs=False, points_on_level_exit=points_on_level_exit,
                                             episode_streams=episode_streams)
        self.torch, self.pool, self.device = torch, pool, base.device
        self.num_envs, self.n_agents = int(num_envs), A
        B = self.num_envs
        base.struct.goal_cache = None       # (the multi-agent kernels keep no goal words: nothing to drop per launch)
        self.view_shape, self.output_channels = base.view_shape, base.output_channels
        self._lib = base._lib
        dev, t = self.device, {}
        L = pool.n_slots
        t["agents"] = torch.zeros((B, A, 12), dtype=torch.int32, device=dev)           # struct sl_agent_state
        la = np.zeros((L, A, 8), np.int32)                                              # struct sl_level_agent
        la[:, :, 0:2] = pool.pool_agent_locs if A > 1 else pool.pool_agent_loc[:, None, :]
        if A > 1:
            la[:, :, 2], la[:, :, 3] = pool.pool_agent_required_reset, pool.pool_agent_required_step
            la[:, :, 4], la[:, :, 5] = pool.pool_agent_initial_points, pool.pool_agent_table_idx
        else:
            la[:, 0, 2], la[:, 0, 3] = pool.pool_required_reset, pool.pool_required_step
            la[:, 0, 4], la[:, 0, 5] = pool.pool_initial_points, pool.pool_table_idx
        t["pool_agents"] = torch.from_numpy(la).to(dev)
        t["out"] = torch.zeros((B, A, 4), dtype=torch.int32, device=dev)               # struct sl_step_out
        vh, vw = self.view_shape
        chans = self.output_channels or ()
        if not with_obs:
            self.obs = None
        elif chans:
            self.obs = torch.zeros((B, A, vh, vw, len(chans)), dtype=torch.uint8, device=dev)
        else:
            self.obs = torch.zeros((B, A, vh, vw), dtype=torch.int32, device=dev)      # uint32 payload
        self.t = t
        m = self.struct = _hip.MultiAgent()
        m.n_agents = A
        m.agents, m.pool_agents, m.out = t["agents"].data_ptr(), t["pool_agents"].data_ptr(), t["out"].data_ptr()
        m.obs = None if self.obs is None else self.obs.data_ptr()
        self._mref = C.byref(m)
        out = t["out"]
        flags = out[:, :, 1:2].view(torch.uint8)                                        # done, success, times_up, pad
        self.reward = out[:, :, 0].view(torch.float32)
        self.done = flags[:, :, 0]
        self.info = {"success": flags[:, :, 1], "times_up": flags[:, :, 2],
                     "episode_reward": out[:, :, 2].view(torch.float32), "episode_length": out[:, :, 3]}

    def reset(self, mask=None):
        """SafeLifeEnv.reset() for every env (or those with mask[e] != 0)."""
        mk = None
        if mask is not None:
            mk = self.torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8)).to(self.device)
        _hip.check(self._lib.slhip_env_reset_multi(self.base._sref, self._mref, None if mk is None else mk.data_ptr(),
                                                   _hip.current_stream_ptr()))
        if mk is not None:
            self.torch.cuda.current_stream().synchronize()      # (the mask tensor is the call's own)
        return self.obs

    def step(self, actions):
        torch = self.torch
        a = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
        if tuple(a.shape) != (self.num_envs, self.n_agents):
            raise ValueError("actions must have shape [num_envs, n_agents]")
        self._actions = a                                       # (alive until the next step)
        _hip.check(self._lib.slhip_env_step_multi(self.base._sref, self._mref, a.data_ptr(), _hip.current_stream_ptr()))
        return self.obs, self.reward, self.done, self.info

    def numpy(self, name):
        """Host copy of a state array: the single-agent env's names for what the env owns (board, goals, rng, num_steps,
        level_idx, episode_idx, goals_static, exit_locs), and per agent [B, A(, 2)]: agent_locs, old_value,
        required_points, initial_points, table_idx, episode_length, episode_reward, is_active, reward, done, success,
        times_up; obs."""
        if name == "obs":
            a = self.obs.cpu().numpy()
            return a.view(np.uint32) if self.output_channels is None else a
        if name == "agent_locs":
            return self.t["agents"][:, :, 0:2].cpu().numpy()
        if name in _hip.AGENT_COLS:
            col = self.t["agents"][:, :, _hip.AGENT_COLS[name]].cpu().numpy()
            return col.view(np.float32) if name == "episode_reward" else col
        if name == "reward":
            return self.reward.cpu().numpy()
        if name == "done":
            return self.done.cpu().numpy()
        if name in ("success", "times_up"):
            return self.info[name].cpu().numpy()
        return self.base.numpy(name)
