"""Scripted on-device sanity policies (tensor in, tensor out), cf. the reference's open-loop grasp demo
`XarmPickAndPlace._run_demo` (/root/reference/gym_xarm/envs/xarm_pick_and_place.py:310-349: move above
the object, close the fingers, lift).  Closed-loop version for the batched env; used as a behavioural
regression metric ("scripted-policy lift rate") and as a usage example of the VecEnv."""
import torch

MAX_STEP = 0.25 * 0.25   # max_vel * dt, xarm_pick_and_place.py:41,28,203


class PickAndLiftPolicy:
    """settle -> rise -> move above the object -> descend to the lowest admissible EEF height (:37) -> close -> lift.
    Closed loop: a phase ends when its target is met (rise: EEF at the travel height; move: EEF within `tol` of the point
    above the object; descend: EEF at the grasp height) or after its step budget.  Two properties of the env shape it
    (per-stage numbers in DESIGN.md 1): the Cartesian target leads the EEF by up to 62.5 mm per env step of 1/60 s and
    the motors follow within the step, i.e. the hand travels at up to ~3 m/s - so the hand first
    RISES from the reset pose (fingers 1 cm above the table) and only then translates, or it bats every object within
    ~10 cm off the table (round 2's script did: 39 % of the envs); and a far object needs more steps than a near one.
    Observations follow _get_obs (:220-248)."""

    def __init__(self, num_envs, device, settle=25, rise=8, move=30, descend=14, close=6, lift_z=0.35, tol=0.004):
        self.t = torch.zeros(num_envs, dtype=torch.long, device=device)       # steps since reset
        self.phase = torch.zeros(num_envs, dtype=torch.long, device=device)   # 0 settle 1 rise 2 move 3 descend 4 close 5 lift
        self.tp = torch.zeros(num_envs, dtype=torch.long, device=device)      # steps spent in the current phase
        self.budget = torch.tensor([settle, rise, move, descend, close, 1 << 30], device=device)
        self.lift_z, self.tol = lift_z, tol
        self.anchor = None

    def reset(self, mask=None):
        for x in (self.t, self.phase, self.tp):
            if mask is None:
                x.zero_()
            else:
                x[mask.bool()] = 0

    def __call__(self, obs):
        o = obs["observation"]
        hand = o[:, 0:3]
        eef = hand + torch.tensor([0.0, 0.0, 0.04], device=o.device)   # hand COM is 0.04 below link_eef for a downward tool
        obj = o[:, 8:11]
        if self.anchor is None:
            self.anchor = obj.clone()
        # phase transitions on the state before acting
        xy_err = (eef[:, 0:2] - self.anchor[:, 0:2]).norm(dim=1)
        met = (self.phase == 1) & (eef[:, 2] > 0.24)
        met |= (self.phase == 2) & (xy_err < self.tol)
        met |= (self.phase == 3) & (eef[:, 2] < 0.153) & (xy_err < 2 * self.tol)
        adv = met | (self.tp >= self.budget[self.phase])
        self.anchor = torch.where(((self.phase <= 1) & adv)[:, None], obj, self.anchor)     # latch the settled object
        self.phase = torch.where(adv, (self.phase + 1).clamp(max=5), self.phase)
        self.tp = torch.where(adv, torch.zeros_like(self.tp), self.tp)
        phase = self.phase
        travel = torch.full_like(eef[:, 2], 0.25)
        z = torch.where(phase <= 2, travel, torch.where(phase >= 5, torch.full_like(travel, self.lift_z), torch.full_like(travel, 0.15)))
        tgt = eef.clone()
        over = phase >= 2
        tgt[:, 0] = torch.where(over, self.anchor[:, 0], eef[:, 0])
        tgt[:, 1] = torch.where(over, self.anchor[:, 1], eef[:, 1])
        tgt[:, 2] = torch.where(phase >= 1, z, eef[:, 2])
        a = torch.zeros(o.shape[0], 4, device=o.device)
        a[:, :3] = ((tgt - eef) / MAX_STEP).clamp(-1, 1)
        a[:, 3] = torch.where(phase >= 4, -1.0, 1.0)
        self.t += 1
        self.tp += 1
        return a


def lift_stages(env, steps=95, lift_height=0.15):
    """PickAndLiftPolicy on `env` (auto_reset off; anything with reset / step / get_state over torch tensors - the HIP env
    or the oracle behind tests' adapter): fraction of envs that ever reach each stage of the script.
      hovered   EEF within 1 cm (xy) of the object at some step
      contact   both fingers in contact with the object (the env's grasp flag) at some step
      raised    object more than 5 cm above its resting height while in contact
      lifted    object above `lift_height` at the last step (= lift_rate)"""
    pol = PickAndLiftPolicy(env.num_envs, env.device)
    obs = env.reset()
    n = env.num_envs
    ever = {k: torch.zeros(n, dtype=torch.bool, device=env.device) for k in ("hovered", "contact", "raised")}
    z0 = None
    for t in range(steps):
        obs, rew, done, info = env.step(pol(obs))
        o = obs["observation"]
        touch = env.get_state()[:, 50] > 0.5
        if t == 24:
            z0 = o[:, 10].clone()               # resting height after the settle phase
        ever["hovered"] |= (o[:, 0:2] - o[:, 8:10]).norm(dim=1) < 0.01
        ever["contact"] |= touch
        if z0 is not None:
            ever["raised"] |= touch & (o[:, 10] > z0 + 0.05)
    out = {k: v.float().mean().item() for k, v in ever.items()}
    out["lifted"] = (obs["achieved_goal"][:, 2] > lift_height).float().mean().item()
    return out


def lift_rate(env, steps=95, lift_height=0.15):
    """fraction of envs whose object ends above `lift_height` under PickAndLiftPolicy (auto_reset off)"""
    return lift_stages(env, steps, lift_height)["lifted"]


class HandoverEzPolicy:
    """Tensor version of the reference's scripted handover controller `XarmHandover.ezpolicy`
    (/root/reference/gym_xarm/envs/xarm_handover.py:404-446): arm 1 reaches, grasps and lifts the stick, arm 2
    reaches and grasps it, then arm 1 lets go.  Observation layout as documented there (:406-418)."""

    def __call__(self, obs):
        o = obs["observation"]
        obj, g1, q1, g2, q2 = o[:, 0:3], o[:, 13:16], o[:, 19], o[:, 21:24], o[:, 27]
        n1, n2 = (obj - g1).norm(dim=1), (obj - g2).norm(dim=1)
        ig1, ig2 = (q1 < 0.25) & (n1 < 0.05), (q2 < 0.25) & (n2 < 0.05)
        d1 = obj - g1 + torch.tensor([-0.07, 0.0, 0.0], device=o.device, dtype=o.dtype)
        d2 = obj - g2 + torch.tensor([0.07, 0.0, 0.0], device=o.device, dtype=o.dtype)
        a = torch.zeros(o.shape[0], 8, device=o.device, dtype=o.dtype)
        a[:, 3] = torch.where(n1 < 0.1, -0.5, 0.5)
        a[:, 7] = torch.where(n2 < 0.1, -0.5, 0.5)
        reach1 = ~ig1
        lift1 = ig1 & ~ig2
        both = ig1 & ig2
        a[:, 0:3] = torch.where(reach1[:, None], d1 / d1.norm(dim=1, keepdim=True), a[:, 0:3])
        a[:, 0:3] = torch.where(lift1[:, None], torch.tensor([0.5, 0.0, 0.5], device=o.device, dtype=o.dtype).expand_as(d1), a[:, 0:3])
        a[:, 4:7] = torch.where(lift1[:, None], d2 / d2.norm(dim=1, keepdim=True), a[:, 4:7])
        a[:, 4] = torch.where(both, torch.full_like(a[:, 4], -0.5), a[:, 4])
        return a


class HandoverReleasePolicy:
    """The reference's ezpolicy with the one step it lacks: RELEASE.  ezpolicy keeps closing gripper 1 for as long as the
    stick is near it (:432-435) and, once both arms hold the stick, only pulls arm 2 back (:446) - the stick changes hands
    only if it is torn out of arm 1's fingers.  Here arm 2 steers at the stick's far end until both of its fingers touch
    it (the env's grasp flag, read from the state), holds for `dwell` steps, then arm 1 opens and backs off while arm 2
    keeps its grip and stays put.  Evidence that the restated physics can complete a hand-over; not part of the reference."""

    def __init__(self, env, dwell=3):
        self.env, self.dwell = env, dwell
        self.ez = HandoverEzPolicy()
        self.count = torch.zeros(env.num_envs, dtype=torch.long, device=env.device)
        self.released = torch.zeros(env.num_envs, dtype=torch.bool, device=env.device)

    def __call__(self, obs):
        a = self.ez(obs)
        st = self.env.get_state()
        t1, t2 = st[:, 70] > 0.5, st[:, 71] > 0.5
        up = obs["achieved_goal"][:, 2] > 0.05
        self.count = torch.where(t1 & t2 & up, self.count + 1, torch.zeros_like(self.count))
        self.released |= self.count >= self.dwell
        r = self.released
        hold2 = t2 & up & ~r
        a[:, 4:7] = torch.where(hold2[:, None], torch.zeros_like(a[:, 4:7]), a[:, 4:7])     # arm 2 stops once it has the stick
        a[:, 7] = torch.where(hold2 | r, torch.full_like(a[:, 7], -1.0), a[:, 7])
        a[:, 0:3] = torch.where(r[:, None], torch.tensor([-0.5, 0.0, 0.3], device=a.device).expand(a.shape[0], 3), a[:, 0:3])
        a[:, 3] = torch.where(r, torch.ones_like(a[:, 3]), a[:, 3])
        a[:, 4:7] = torch.where(r[:, None], torch.zeros_like(a[:, 4:7]), a[:, 4:7])
        return a


def handover_stages(env, steps=40, policy=None):
    """HandoverEzPolicy (the reference's own controller) on `env` (auto_reset off; the HIP env or the oracle behind tests'
    adapter): of the envs whose stick starts on arm 1's side, the fraction that ever reach each stage the policy is
    written to produce (xarm_handover.py:419-446):
      reached    stick within 0.1 of arm 1's grip point (where the policy starts closing, :432)
      grasp1     both fingers of arm 1 in contact with the stick (the env's grasp flag, :263)
      lifted     ... while the stick is more than 5 cm up (arm 1 carries it, :440)
      contact2   both fingers of arm 2 in contact while it is up (mutual grasp, stage 5 of :157-163)
      handed     arm 2 alone holds it above 6 cm at some step (= handover_rate)
      held_end   ... and still does at the last step"""
    pol = policy if policy is not None else HandoverEzPolicy()
    obs = env.reset()
    x0 = obs["achieved_goal"][:, 0].clone()
    mine = x0 < 0
    n = env.num_envs
    ever = {k: torch.zeros(n, dtype=torch.bool, device=env.device) for k in ("reached", "grasp1", "lifted", "contact2", "handed")}
    held = torch.zeros(n, dtype=torch.bool, device=env.device)
    for _ in range(steps):
        obs, rew, done, info = env.step(pol(obs))
        st = env.get_state()
        o = obs["observation"]
        t1, t2, z = st[:, 70] > 0.5, st[:, 71] > 0.5, obs["achieved_goal"][:, 2]
        ever["reached"] |= (o[:, 0:3] - o[:, 13:16]).norm(dim=1) < 0.1
        ever["grasp1"] |= t1
        ever["lifted"] |= t1 & (z > 0.05)
        ever["contact2"] |= t2 & (z > 0.05)
        held = ~t1 & t2 & (z > 0.06)
        ever["handed"] |= held
    den = max(mine.float().sum().item(), 1.0)
    out = {k: (v & mine).float().sum().item() / den for k, v in ever.items()}
    out["held_end"] = (held & mine).float().sum().item() / den
    return out


def handover_rate(env, steps=40):
    """fraction of envs (auto_reset off) in which, under HandoverEzPolicy, the stick is lifted and held by arm 2
    alone at some step - the event the reference's policy is written to produce"""
    return handover_stages(env, steps)["handed"]
