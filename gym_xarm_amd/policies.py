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
