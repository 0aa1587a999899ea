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
    """

    def __init__(self, env, policy, generator=None, copy_obs=True):
        import torch
        self.torch = torch
        if env.policy_tensor is None:
            raise ValueError("VectorRunner needs SafeLifeVectorEnv(policy_layout=...)")
        if not env.auto_reset:
            raise ValueError("VectorRunner needs auto_reset=True (finished envs reload inside the step kernel)")
        self.env, self.policy, self.generator, self.copy_obs = env, policy, generator, copy_obs
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
        model_in = obs if obs.dtype == torch.float32 else obs.to(torch.float32)
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
