"""Hindsight-experience-replay buffer for the batched goal envs, resident on the env's device.

The reference's training script asks SB3 for `HerReplayBuffer(n_sampled_goal=4, goal_selection_strategy="future",
online_sampling=True)` (benchmark/train.py:82-97); what that buffer needs from the env is exactly the batched
`compute_reward(achieved_goal, goal, info)` of the GoalEnv API (xarm_pick_and_place.py:155-177, xarm_reach.py:107-116,
xarm_handover.py:153-183).  Here the transitions of all E envs are stored time-major as device tensors and a
sampled batch is relabelled with one gather + one `compute_reward` kernel launch (the C-ABI's `xarm_compute_reward`),
no host round trip.

Layout: a ring of `horizon` time slots, each holding one transition per env.  `ep_end[slot, env]` is the absolute
time of the last transition of the episode the entry belongs to, or -1 while that episode is still running; only
entries of finished episodes are sampled ("future" needs the episode's end).
"""
import torch


class HerReplayBuffer:
    def __init__(self, env, horizon=None, n_sampled_goal=4, goal_selection_strategy="future", reward_fn=None, seed=0):
        assert goal_selection_strategy in ("future", "final"), goal_selection_strategy
        self.E, self.device = env.num_envs, env.device
        max_len = int(getattr(env, "max_episode_steps", 50))
        self.horizon = int(horizon) if horizon else 4 * max_len
        assert self.horizon >= 2 * max_len, "the ring must hold at least two full episodes per env"
        self.strategy = goal_selection_strategy
        self.her_ratio = 1.0 - 1.0 / (n_sampled_goal + 1)          # SB3: n_sampled_goal virtual per real transition
        self.reward_fn = reward_fn if reward_fn is not None else env.compute_reward
        T, E, dev = self.horizon, self.E, self.device
        f = dict(device=dev, dtype=torch.float32)
        self.obs = torch.zeros(T, E, env.obs_dim, **f)
        self.next_obs = torch.zeros(T, E, env.obs_dim, **f)
        self.ag = torch.zeros(T, E, env.goal_dim, **f)
        self.next_ag = torch.zeros(T, E, env.goal_dim, **f)
        self.dg = torch.zeros(T, E, env.goal_dim, **f)
        self.act = torch.zeros(T, E, env.action_dim, **f)
        self.rew = torch.zeros(T, E, **f)
        self.done = torch.zeros(T, E, device=dev, dtype=torch.bool)
        self.ep_end = torch.full((T, E), -1, device=dev, dtype=torch.int64)
        self.slot_time = torch.full((T,), -1, device=dev, dtype=torch.int64)   # absolute time stored in each slot
        self.ep_start = torch.zeros(E, device=dev, dtype=torch.int64)          # absolute time each env's episode began
        self.t = 0
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)

    def add(self, obs, next_obs, action, reward, done):
        """One transition per env.  `obs` / `next_obs` are the env's dicts; for envs with `done` set pass the
        terminal observation (info["terminal_observation"]) as `next_obs`, not the post-reset one."""
        s = self.t % self.horizon
        self.obs[s], self.next_obs[s] = obs["observation"], next_obs["observation"]
        self.ag[s], self.next_ag[s], self.dg[s] = obs["achieved_goal"], next_obs["achieved_goal"], obs["desired_goal"]
        self.act[s], self.rew[s] = action, reward
        d = done.to(torch.bool)
        self.done[s] = d
        self.ep_end[s] = -1
        self.slot_time[s] = self.t
        # close the episodes that ended now: every stored entry of env e with ep_start[e] <= time <= t
        inside = (self.slot_time[:, None] >= self.ep_start[None, :]) & d[None, :]
        self.ep_end = torch.where(inside, torch.full_like(self.ep_end, self.t), self.ep_end)
        self.ep_start = torch.where(d, torch.full_like(self.ep_start, self.t + 1), self.ep_start)
        self.t += 1

    def num_valid(self):
        return int(self._valid().sum().item())

    def _valid(self):
        # finished episodes whose first entry has not been overwritten since
        return (self.ep_end >= 0) & (self.slot_time[:, None] > self.t - 1 - self.horizon)

    def sample(self, batch_size):
        """-> dict of [B, .] tensors; the first round(her_ratio * B) rows carry a relabelled goal and a recomputed
        reward, the rest are the stored transitions."""
        valid = self._valid().flatten().nonzero().squeeze(1)
        if valid.numel() == 0:
            raise RuntimeError("HerReplayBuffer.sample: no finished episode stored yet")
        pick = valid[torch.randint(valid.numel(), (batch_size,), device=self.device, generator=self.gen)]
        s, e = pick // self.E, pick % self.E
        t_abs, end = self.slot_time[s], self.ep_end[s, e]
        n_her = int(round(self.her_ratio * batch_size))
        if self.strategy == "future":
            u = torch.rand(batch_size, device=self.device, generator=self.gen)
            t_goal = t_abs + (u * (end - t_abs + 1).to(torch.float32)).to(torch.int64)
            t_goal = torch.minimum(t_goal, end)
        else:
            t_goal = end
        new_goal = self.next_ag[t_goal % self.horizon, e]
        goal = self.dg[s, e].clone()
        goal[:n_her] = new_goal[:n_her]
        next_ag = self.next_ag[s, e]
        reward = self.rew[s, e].clone()
        if n_her > 0:
            reward[:n_her] = self.reward_fn(next_ag[:n_her].contiguous(), goal[:n_her].contiguous(), None)
        return {"observation": self.obs[s, e], "next_observation": self.next_obs[s, e], "achieved_goal": self.ag[s, e],
                "next_achieved_goal": next_ag, "desired_goal": goal, "action": self.act[s, e], "reward": reward,
                "done": self.done[s, e], "relabelled": torch.arange(batch_size, device=self.device) < n_her,
                "env": e, "time": t_abs, "goal_time": t_goal}


def collect(env, buffer, policy, steps, obs=None):
    """Roll `steps` env steps with policy(obs_dict) -> actions[E, A] and store them.  The env's tensors are reused
    from step to step, so the previous observation is cloned before stepping, and finished envs contribute their
    terminal observation (info['terminal_observation']) instead of the post-reset one."""
    assert not getattr(env, "_lazy", False), "HER collection assumes the reference's in-call auto-reset (auto_reset=True)"
    if obs is None:
        obs = env.reset()
    for _ in range(steps):
        prev = {k: v.clone() for k, v in obs.items()}
        act = policy(prev)
        obs, rew, done, info = env.step(act)
        d = (done != 0)[:, None]
        term = info["terminal_observation"]
        nxt = {"observation": torch.where(d, term, obs["observation"]),
               "achieved_goal": torch.where(d, env.achieved_goal_of(term), obs["achieved_goal"])}
        buffer.add(prev, nxt, act, rew, done)
    return obs
