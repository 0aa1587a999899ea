"""
``VectorRunner`` -- the driver loop of the reference's trainers, batched and device-resident.

The reference steps a Python list of environments one by one: ``obs_for_envs`` collects the observations of
the active agents, the algorithm's ``take_one_step`` runs the model on them and samples an action per agent
with numpy on the host, ``act_on_envs`` steps every env, resets the ones that finished and remembers their
last observation (training/base_algo.py:152-244, training/ppo.py:61-73).  Here the envs are one
``SafeLifeVectorEnv``: the observation arrives from the step kernel already in the layout the policy network
convolves (``policy_layout``), actions are sampled on the device, finished envs are reloaded inside the step
kernel (``auto_reset``), and nothing visits the host.

What carries over from the reference, per step and per agent (= per env: the fused path is single-agent):
``obs, actions, rewards, done, next_obs, agent_ids, policies, values`` with the reference's meaning --
``done`` describes the step that was just taken, ``next_obs`` of a finished env is already the first
observation of its next episode, and an agent id changes when its env resets (``(env index, resets so far)``
instead of ``(id(env), env.num_resets, k)``), so trajectories are strung together exactly as
``gen_training_batch`` does.
"""
import collections

StepResult = collections.namedtuple("StepResult", "obs actions rewards done next_obs agent_ids policies values")


class VectorRunner(object):
    """
    Parameters
    ----------
    env : SafeLifeVectorEnv      built with ``policy_layout="float32"`` (or "uint8") and ``auto_reset=True``
    policy : callable            ``policy(obs [B,C,W,H]) -> (values [B], action probabilities [B,9])`` on the
                                 device -- the interface of ``SafeLifePolicyNetwork.forward`` after its transpose
                                 (training/models.py:99-109); a torch module or any function
    generator : torch.Generator  for the action draws (device generator); None = torch's default
    copy_obs : bool              ``obs`` / ``next_obs`` of a result are views of the env's tensor, which the next
                                 step overwrites; True returns an own copy of ``obs`` (what a replay buffer needs)
    cast_obs : bool              True: the policy gets float32 (training/ppo.py:64); False: the env's tensor as it is
    """

    def __init__(self, env, policy, generator=None, copy_obs=True, cast_obs=True):
        import torch
        self.torch = torch
        if env.policy_tensor is None:
            raise ValueError("VectorRunner needs SafeLifeVectorEnv(policy_layout=...)")
        if not env.auto_reset:
            raise ValueError("VectorRunner needs auto_reset=True (finished envs reload inside the step kernel)")
        self.env, self.policy, self.generator, self.copy_obs, self.cast_obs = env, policy, generator, copy_obs, cast_obs
        B = env.num_envs
        self.env_ids = torch.arange(B, device=env.device, dtype=torch.int64)
        self.num_resets = torch.zeros(B, device=env.device, dtype=torch.int64)       # env.num_resets of the reference
        self.num_steps = 0
        self._started = False

    def obs_for_envs(self):
        """Current observation of every env and its agent id ``(env index, resets so far)``; the first call
        resets the envs (training/base_algo.py:166-171)."""
        if not self._started:
            self.env.reset()
            self._started = True
        return self.env.policy_tensor, (self.env_ids, self.num_resets.clone())

    def act_on_envs(self, actions):
        """Step every env with its action; returns ``(next_obs, rewards, done)`` -- ``rewards`` is the wrapped
        (float64) reward when the env was built with ``wrappers=``, the game's float32 reward otherwise; the raw one
        stays available as ``env.reward``.  An env whose episode ended has
        already been reset: its ``next_obs`` row is the new episode's first observation and its reset counter is
        bumped (training/base_algo.py:231-238)."""
        torch = self.torch
        a = actions.to(device=self.env.device, dtype=torch.int32).contiguous()
        self.env.step(a)
        # what the reference's trainers see is the reward as the wrapper stack hands it on (movement bonus, exit
        # bonus, side-effect penalty: env_factory.py:277-283 wraps the env before base_algo steps it)
        shaped = getattr(self.env, "shaped_reward", None)
        rewards = shaped.clone() if shaped is not None else self.env.reward.clone()
        done = self.env.done.to(torch.bool)
        self.num_resets += done.to(torch.int64)
        self.num_steps += 1
        return self.env.policy_tensor, rewards, done

    def take_one_step(self):
        """training/ppo.py:61-73 without the host: model forward, one categorical draw per env on the device,
        the fused step."""
        torch = self.torch
        obs, agent_ids = self.obs_for_envs()
        # (training/ppo.py:64 hands the network float32; cast_obs=False leaves a uint8 policy tensor as it is)
        model_in = obs if (obs.dtype == torch.float32 or not self.cast_obs) else obs.to(torch.float32)
        with torch.no_grad():
            values, policies = self.policy(model_in)
        actions = torch.multinomial(policies, 1, generator=self.generator).squeeze(1)
        kept = obs.clone() if self.copy_obs else obs
        next_obs, rewards, done = self.act_on_envs(actions)
        return StepResult(kept, actions, rewards, done, next_obs, agent_ids, policies, values)

    def run_steps(self, n):
        """n steps; yields each StepResult (a generator, so that a learner can consume them as they come)."""
        for _ in range(n):
            yield self.take_one_step()


class PipelinedRunner(object):
    """The same loop with the envs in GROUPS (the env's slices), each on a stream of its own: a group's observation ->
    policy -> action draw -> step all run on that stream, in that order, so nothing inside a group needs a fence, and the
    groups overlap -- one group's step kernel runs while another group's policy does (the reference's trainers walk their
    envs one after the other, training/base_algo.py:208-238; here the walk is over groups and the device does two things
    at once).  The draw is the library's own kernel (``slhip_sample_actions``: one thread per env, straight into the
    group's part of the ONE int32 action tensor the step kernel reads); ``sampler="torch"`` uses ``torch.multinomial``
    and a copy instead (20 times the device time at 8192 envs).

    Parameters: `env` built with ``policy_layout=...``, ``auto_reset=True``, ``slices >= 2``.
    ``policy(obs [n,C,W,H]) -> (values, probs float32 [n,9])`` gets the env's policy tensor AS IT IS (uint8 or
    float32: a network casts its own input, a cheap policy need not pay for a float copy of the observation).
    ``on_step(group, lo, hi)`` (optional) is called, with the group's stream current, after each group step: the
    group's ``env.reward[lo:hi]`` / ``env.done[lo:hi]`` / ``env.policy_tensor[lo:hi]`` are valid on that stream there.
    """

    def __init__(self, env, policy, seed=0, on_step=None, sampler="device", generator=None):
        import torch
        from . import _hip
        self.torch, self._hip = torch, _hip
        if env.policy_tensor is None or not env.auto_reset or env.slices < 2:
            raise ValueError("PipelinedRunner needs SafeLifeVectorEnv(policy_layout=..., auto_reset=True, slices>=2)")
        if sampler not in ("device", "torch"):
            raise ValueError("sampler must be 'device' or 'torch'")
        self.env, self.policy, self.on_step, self.sampler, self.generator = env, policy, on_step, sampler, generator
        self.seed = int(seed) & (2 ** 64 - 1)
        self.actions = torch.zeros(env.num_envs, dtype=torch.int32, device=env.device)
        self.num_steps = 0
        self._group_steps = [0] * env.slices        # steps taken per group: the draw counter of the group's next step
        self._started = False
        self._lib = _hip.lib()
        self._groups = []
        for g in range(env.slices):
            lo, hi = env.slice_bounds[g], env.slice_bounds[g + 1]
            st = env.slice_stream(g)
            self._groups.append((g, lo, hi, st, torch.cuda.stream(st), env.policy_tensor[lo:hi],
                                 self.actions.data_ptr() + 4 * lo, st.cuda_stream))

    def start(self):
        if not self._started:
            self.env.reset()
            self.env.fence()                    # the groups' streams wait for the reset once
            self._started = True

    def step_group(self, g):
        torch, env = self.torch, self.env
        g, lo, hi, st, ctx, obs, act_ptr, st_ptr = self._groups[g]
        if hi <= lo:
            return
        # Work the caller put on ITS stream since the last fence (env.reset(mask), step(), a snapshot ...) is fenced HERE,
        # while the caller's stream is still the current one: inside the group's stream context env.fence() would take
        # the group's own stream for the caller's and order the slices against the wrong one.
        if env._queues_pending:
            env._settle()
        if env._caller_ahead:
            env.fence()
        draw = self._group_steps[g]
        self._group_steps[g] = draw + 1
        with ctx:
            with torch.no_grad():
                values, probs = self.policy(obs)
            if self.sampler == "device":
                if probs.dtype != torch.float32 or not probs.is_contiguous():
                    probs = probs.to(torch.float32).contiguous()
                # (seed offset by the group's first env: every env of the batch has its own draw per step)
                rc = self._lib.slhip_sample_actions(probs.data_ptr(), hi - lo, probs.shape[1], (self.seed + lo) & (2 ** 64 - 1),
                                                    draw, act_ptr, st_ptr)
                if rc:
                    self._hip.check(rc)
            else:
                drawn = torch.multinomial(probs, 1, generator=self.generator)
                self.actions[lo:hi].copy_(drawn.view(-1))       # int64 -> int32, into the step's buffer
            env.step_slice(g, self.actions)
            if self.on_step is not None:
                self.on_step(g, lo, hi)

    def run(self, n_steps):
        """n_steps steps of every env."""
        self.start()
        for _ in range(n_steps):
            for g in range(self.env.slices):
                self.step_group(g)
            self.num_steps = min(self._group_steps)

    def finish(self):
        """The caller's current stream waits for every group."""
        self.env.join()
