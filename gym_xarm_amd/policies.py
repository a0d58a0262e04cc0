"""Scripted on-device sanity policies (tensor in, tensor out), cf. the reference's open-loop grasp demo
`XarmPickAndPlace._run_demo` (/root/reference/gym_xarm/envs/xarm_pick_and_place.py:310-349: move above
the object, close the fingers, lift).  Closed-loop version for the batched env; used as a behavioural
regression metric ("scripted-policy lift rate") and as a usage example of the VecEnv."""
import torch

MAX_STEP = 0.25 * 0.25   # max_vel * dt, xarm_pick_and_place.py:41,28,203


class PickAndLiftPolicy:
    """settle -> hover above the object -> descend to the lowest admissible EEF height (:37) -> close ->
    lift.  Phases advance per env on a step counter, observations follow _get_obs (:220-248)."""

    def __init__(self, num_envs, device, settle=25, hover=6, descend=8, close=6, lift_z=0.35):
        self.t = torch.zeros(num_envs, dtype=torch.long, device=device)
        self.bounds = torch.tensor([settle, settle + hover, settle + hover + descend, settle + hover + descend + close], device=device)
        self.lift_z = lift_z
        self.anchor = None

    def reset(self, mask=None):
        if mask is None:
            self.t.zero_()
        else:
            self.t[mask.bool()] = 0

    def __call__(self, obs):
        o = obs["observation"]
        hand = o[:, 0:3]
        eef = hand + torch.tensor([0.0, 0.0, 0.04], device=o.device)   # hand COM is 0.04 below link_eef for a downward tool
        obj = o[:, 8:11]
        phase = torch.bucketize(self.t, self.bounds, right=True)
        if self.anchor is None:
            self.anchor = obj.clone()
        fresh = self.t == self.bounds[0]
        self.anchor = torch.where(fresh[:, None], obj, self.anchor)
        tgt = eef.clone()
        z = torch.where(phase == 1, torch.full_like(eef[:, 2], 0.25), torch.where(phase >= 4, torch.full_like(eef[:, 2], self.lift_z),
                                                                                  torch.full_like(eef[:, 2], 0.15)))
        moving = phase >= 1
        tgt[:, 0] = torch.where(moving, self.anchor[:, 0], eef[:, 0])
        tgt[:, 1] = torch.where(moving, self.anchor[:, 1], eef[:, 1])
        tgt[:, 2] = torch.where(moving, z, eef[:, 2])
        a = torch.zeros(o.shape[0], 4, device=o.device)
        a[:, :3] = ((tgt - eef) / MAX_STEP).clamp(-1, 1)
        a[:, 3] = torch.where(phase >= 3, -1.0, 1.0)
        self.t += 1
        return a


def lift_rate(env, steps=58, lift_height=0.15):
    """fraction of envs whose object ends above `lift_height` under PickAndLiftPolicy (auto_reset off)"""
    pol = PickAndLiftPolicy(env.num_envs, env.device)
    obs = env.reset()
    for _ in range(steps):
        obs, rew, done, info = env.step(pol(obs))
    return (obs["achieved_goal"][:, 2] > lift_height).float().mean().item()


class HandoverEzPolicy:
    """Tensor version of the reference's scripted handover controller `XarmHandover.ezpolicy`
    (/root/reference/gym_xarm/envs/xarm_handover.py:404-446): arm 1 reaches, grasps and lifts the stick, arm 2
    reaches and grasps it, then arm 1 lets go.  Observation layout as documented there (:406-418)."""

    def __call__(self, obs):
        o = obs["observation"]
        obj, g1, q1, g2, q2 = o[:, 0:3], o[:, 13:16], o[:, 19], o[:, 21:24], o[:, 27]
        n1, n2 = (obj - g1).norm(dim=1), (obj - g2).norm(dim=1)
        ig1, ig2 = (q1 < 0.25) & (n1 < 0.05), (q2 < 0.25) & (n2 < 0.05)
        d1 = obj - g1 + torch.tensor([-0.07, 0.0, 0.0], device=o.device)
        d2 = obj - g2 + torch.tensor([0.07, 0.0, 0.0], device=o.device)
        a = torch.zeros(o.shape[0], 8, device=o.device)
        a[:, 3] = torch.where(n1 < 0.1, -0.5, 0.5)
        a[:, 7] = torch.where(n2 < 0.1, -0.5, 0.5)
        reach1 = ~ig1
        lift1 = ig1 & ~ig2
        both = ig1 & ig2
        a[:, 0:3] = torch.where(reach1[:, None], d1 / d1.norm(dim=1, keepdim=True), a[:, 0:3])
        a[:, 0:3] = torch.where(lift1[:, None], torch.tensor([0.5, 0.0, 0.5], device=o.device).expand_as(d1), a[:, 0:3])
        a[:, 4:7] = torch.where(lift1[:, None], d2 / d2.norm(dim=1, keepdim=True), a[:, 4:7])
        a[:, 4] = torch.where(both, torch.full_like(a[:, 4], -0.5), a[:, 4])
        return a


def handover_rate(env, steps=40):
    """fraction of envs (auto_reset off) in which, under HandoverEzPolicy, the stick is lifted and held by arm 2
    alone at some step - the event the reference's policy is written to produce"""
    pol = HandoverEzPolicy()
    obs = env.reset()
    x0 = obs["achieved_goal"][:, 0].clone()
    handed = torch.zeros(env.num_envs, dtype=torch.bool, device=env.device)
    for _ in range(steps):
        obs, rew, done, info = env.step(pol(obs))
        st = env.get_state()
        handed |= (st[:, 70] < 0.5) & (st[:, 71] > 0.5) & (obs["achieved_goal"][:, 2] > 0.06) & (x0 < 0)
    return handed.float().sum().item() / max((x0 < 0).float().sum().item(), 1.0)
