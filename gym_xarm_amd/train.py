"""On-device A2C training driver for the batched envs - the MI355X counterpart of the reference's
`benchmark/train.py:65-111` (SB3: make_vec_env -> VecNormalize(norm_obs, norm_reward, clip_obs=10) -> A2C
'MlpPolicy' -> learn -> save).  Everything (env step, normalisation statistics, policy, optimiser) stays on the
GPU; there is no host round trip per step.  SB3 itself is not installed here, so the A2C update (n_steps = 5,
gamma 0.99, GAE lambda 1, value coefficient 0.5, entropy coefficient 0, RMSprop 7e-4 - SB3's defaults) is written
out in ~60 lines of torch.

  python -m gym_xarm_amd.train --env XarmReach-v0 --num-envs 4096 --updates 300 --reward-type dense
"""
import argparse
import json
import time

import torch
import torch.nn as nn


class RunningMeanStd:
    """VecNormalize's running statistics (parallel-variance update), as tensors on the env's device"""

    def __init__(self, shape, device):
        self.mean = torch.zeros(shape, device=device)
        self.var = torch.ones(shape, device=device)
        self.count = 1e-4

    def update(self, x):
        b_mean, b_var, b_n = x.mean(0), x.var(0, unbiased=False), x.shape[0]
        delta, tot = b_mean - self.mean, self.count + b_n
        self.mean = self.mean + delta * b_n / tot
        self.var = (self.var * self.count + b_var * b_n + delta ** 2 * self.count * b_n / tot) / tot
        self.count = tot


class VecNormalize:
    """obs / reward normalisation wrapper (benchmark/train.py:75: norm_obs=True, norm_reward=True, clip_obs=10)"""

    def __init__(self, env, clip_obs=10.0, clip_reward=10.0, gamma=0.99, eps=1e-8):
        self.env, self.clip_obs, self.clip_reward, self.gamma, self.eps = env, clip_obs, clip_reward, gamma, eps
        dev = env.device
        self.dim = env.obs_dim + 2 * env.goal_dim
        self.obs_rms = RunningMeanStd((self.dim,), dev)
        self.ret_rms = RunningMeanStd((), dev)
        self.ret = torch.zeros(env.num_envs, device=dev)
        self.training = True

    def _flat(self, obs):
        return torch.cat([obs["observation"], obs["achieved_goal"], obs["desired_goal"]], dim=1)

    def _norm(self, x):
        if self.training:
            self.obs_rms.update(x)
        return ((x - self.obs_rms.mean) / torch.sqrt(self.obs_rms.var + self.eps)).clamp(-self.clip_obs, self.clip_obs)

    def reset(self):
        self.ret.zero_()
        return self._norm(self._flat(self.env.reset()))

    def step(self, actions):
        obs, rew, done, info = self.env.step(actions)
        self.ret = self.ret * self.gamma + rew
        if self.training:
            self.ret_rms.update(self.ret)
        nrew = (rew / torch.sqrt(self.ret_rms.var + self.eps)).clamp(-self.clip_reward, self.clip_reward)
        self.ret = torch.where(done != 0, torch.zeros_like(self.ret), self.ret)
        return self._norm(self._flat(obs)), nrew, done, info, rew

    def state_dict(self):
        return {"obs_mean": self.obs_rms.mean, "obs_var": self.obs_rms.var, "obs_count": torch.tensor(self.obs_rms.count),
                "ret_var": self.ret_rms.var, "ret_count": torch.tensor(self.ret_rms.count)}


class ActorCritic(nn.Module):
    """SB3 'MlpPolicy': separate 64-64 tanh towers for policy and value, state-independent log-std"""

    def __init__(self, obs_dim, act_dim):
        super().__init__()
        def tower(out):
            return nn.Sequential(nn.Linear(obs_dim, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh(), nn.Linear(64, out))
        self.pi, self.vf = tower(act_dim), tower(1)
        self.log_std = nn.Parameter(torch.zeros(act_dim))

    def dist(self, x):
        return torch.distributions.Normal(self.pi(x), self.log_std.exp())

    def value(self, x):
        return self.vf(x).squeeze(-1)


def train(env_id="XarmReach-v0", num_envs=4096, updates=300, n_steps=5, gamma=0.99, lr=7e-4, seed=0, config=None, log_every=50,
          quiet=False, auto_reset=True):
    """auto_reset="lazy" (PickAndPlace): transitions flagged info["resetting"] carry no reward and no gradient and cut the
    return like an episode end - the env spends them on its reset ticks (include/xarm_hip.h XARM_AUTO_RESET_LAZY)"""
    import gym_xarm_amd
    torch.manual_seed(seed)
    env = gym_xarm_amd.make(env_id, num_envs=num_envs, seed=seed, config=config, auto_reset=auto_reset)
    venv = VecNormalize(env, gamma=gamma)
    dev = env.device
    model = ActorCritic(venv.dim, env.act_dim).to(dev)
    opt = torch.optim.RMSprop(model.parameters(), lr=lr, alpha=0.99, eps=1e-5)
    obs = venv.reset()
    hist, t0 = [], time.perf_counter()
    succ_sum, done_sum, raw_sum = torch.zeros((), device=dev), torch.zeros((), device=dev), torch.zeros((), device=dev)
    for it in range(1, updates + 1):
        obs_buf, act_buf, rew_buf, done_buf, use_buf = [], [], [], [], []
        for _ in range(n_steps):
            with torch.no_grad():
                a = model.dist(obs).sample()
            nobs, nrew, done, info, raw = venv.step(a.clamp(-1, 1))
            resetting = info["resetting"].float() if "resetting" in info else torch.zeros_like(nrew)
            obs_buf.append(obs); act_buf.append(a); rew_buf.append(nrew * (1.0 - resetting))
            done_buf.append(torch.maximum(done.float(), resetting)); use_buf.append(1.0 - resetting)
            succ_sum += (info["is_success"].float() * done.float()).sum()
            done_sum += done.float().sum()
            raw_sum += raw.mean()
            obs = nobs
        with torch.no_grad():
            ret = model.value(obs)
            rets = []
            for k in reversed(range(n_steps)):
                ret = rew_buf[k] + gamma * ret * (1.0 - done_buf[k])
                rets.append(ret)
            rets = torch.stack(rets[::-1])
        O, A, U = torch.stack(obs_buf), torch.stack(act_buf), torch.stack(use_buf)
        values = model.value(O)
        adv = rets - values.detach()
        logp = model.dist(O).log_prob(A).sum(-1)
        n_use = U.sum().clamp(min=1.0)
        loss = -(adv * logp * U).sum() / n_use + 0.5 * (((rets - values) ** 2) * U).sum() / n_use
        opt.zero_grad(set_to_none=True)
        loss.backward()
        nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        if it % log_every == 0 or it == updates:
            torch.cuda.synchronize()
            rec = {"update": it, "env_steps": it * n_steps * num_envs, "mean_raw_reward": (raw_sum / (log_every * n_steps)).item(),
                   "success_rate": (succ_sum / done_sum.clamp(min=1)).item(), "episodes": int(done_sum.item()),
                   "env_steps_per_sec": it * n_steps * num_envs / (time.perf_counter() - t0)}
            hist.append(rec)
            if not quiet:
                print(json.dumps(rec), flush=True)
            succ_sum.zero_(); done_sum.zero_(); raw_sum.zero_()
    env.close()
    return model, venv, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="XarmReach-v0")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--updates", type=int, default=300)
    ap.add_argument("--reward-type", default="dense")
    ap.add_argument("--save", default=None, help="safetensors file for the policy + VecNormalize statistics")
    ap.add_argument("--lazy-reset", action="store_true", help="opt-in lazy auto-reset (PickAndPlace), masked in the update")
    args = ap.parse_args()
    cfg = {"reward_type": args.reward_type, "GUI": False} if ("Reach" in args.env or "PickAndPlace" in args.env) else None
    model, venv, hist = train(args.env, args.num_envs, args.updates, config=cfg, auto_reset="lazy" if args.lazy_reset else True)
    if args.save:
        from safetensors.torch import save_file
        sd = {"policy." + k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
        sd.update({"vecnormalize." + k: v.detach().cpu().contiguous() for k, v in venv.state_dict().items()})
        save_file(sd, args.save)


if __name__ == "__main__":
    main()
