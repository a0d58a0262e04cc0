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
import os
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
        # a flat-observation ('NoGoal') env hands out the observation tensor itself (gym_xarm_amd/sb3_adapter.py FlatObsVecEnv)
        self.flat = bool(getattr(env, "flat_observation", False))
        self.dim = env.obs_dim if self.flat else env.obs_dim + 2 * env.goal_dim
        self.obs_rms = RunningMeanStd((self.dim,), dev)
        self.ret_rms = RunningMeanStd((), dev)
        self.ret = torch.zeros(env.num_envs, device=dev)
        self.training = True

    def _flat(self, obs):
        if self.flat:
            return obs
        return torch.cat([obs["observation"], obs["achieved_goal"], obs["desired_goal"]], dim=1)

    def _norm(self, x, keep=None):
        """keep: bool [E] rows that feed the running statistics (None = all)"""
        if self.training:
            rows = x if keep is None else x[keep]
            if rows.shape[0] > 0:
                self.obs_rms.update(rows)
        return ((x - self.obs_rms.mean) / torch.sqrt(self.obs_rms.var + self.eps)).clamp(-self.clip_obs, self.clip_obs)

    def reset(self):
        self.ret.zero_()
        return self._norm(self._flat(self.env.reset()))

    def step(self, actions):
        obs, rew, done, info = self.env.step(actions)
        # lazy auto-reset: rows that are spending this call on a reset tick (info['resetting']) are not transitions of
        # the task - their observation (desired_goal holds scratch values then) and reward stay out of the statistics
        keep = ~info["resetting"] if "resetting" in info else None
        self.ret = self.ret * self.gamma + rew
        if self.training:
            r = self.ret if keep is None else self.ret[keep]
            if r.shape[0] > 0:
                self.ret_rms.update(r)
        nrew = (rew / torch.sqrt(self.ret_rms.var + self.eps)).clamp(-self.clip_reward, self.clip_reward)
        self.ret = torch.where(done != 0, torch.zeros_like(self.ret), self.ret)
        return self._norm(self._flat(obs), keep), nrew, done, info, rew

    def state_dict(self):
        return {"obs_mean": self.obs_rms.mean, "obs_var": self.obs_rms.var, "obs_count": torch.tensor(float(self.obs_rms.count), dtype=torch.float64),
                "ret_mean": self.ret_rms.mean, "ret_var": self.ret_rms.var, "ret_count": torch.tensor(float(self.ret_rms.count), dtype=torch.float64),
                "clip_obs": torch.tensor(float(self.clip_obs)), "clip_reward": torch.tensor(float(self.clip_reward)),
                "gamma": torch.tensor(float(self.gamma))}

    def load_state_dict(self, sd):
        dev = self.obs_rms.mean.device
        if tuple(sd["obs_mean"].shape) != tuple(self.obs_rms.mean.shape):
            raise ValueError("VecNormalize statistics are for observation width %d, this env has %d" % (sd["obs_mean"].shape[0], self.dim))
        self.obs_rms.mean, self.obs_rms.var = sd["obs_mean"].to(dev).clone(), sd["obs_var"].to(dev).clone()
        self.obs_rms.count = float(sd["obs_count"])
        # files written before the return statistics / clip settings were saved (round 1's --save) hold the observation
        # statistics only: keep this wrapper's own values for what the file lacks
        if "ret_mean" in sd:
            self.ret_rms.mean, self.ret_rms.var = sd["ret_mean"].to(dev).clone(), sd["ret_var"].to(dev).clone()
            self.ret_rms.count = float(sd["ret_count"])
        self.clip_obs = float(sd["clip_obs"]) if "clip_obs" in sd else self.clip_obs
        self.clip_reward = float(sd["clip_reward"]) if "clip_reward" in sd else self.clip_reward
        self.gamma = float(sd["gamma"]) if "gamma" in sd else self.gamma

    def save(self, path):
        """counterpart of `env.save(stats_path)` (benchmark/train.py:107-108): the running statistics and the clip /
        discount settings in one safetensors file (the reference pickles the wrapper; nothing is executed on load here)"""
        from safetensors.torch import save_file
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, path)

    @classmethod
    def load(cls, path, env):
        """counterpart of `VecNormalize.load(stats_path, env)` (benchmark/display.py:22)"""
        from safetensors.torch import load_file
        v = cls(env)
        v.load_state_dict(load_file(path))
        return v


class EpisodeMonitor:
    """SB3 `Monitor` for a device-resident batch (benchmark/train.py:74 `monitor_dir`): per-env return and length of
    the running episode are accumulated on the device; every finished episode appends one `r,l,t` row (raw reward sum,
    length, seconds since start) to a device ring, flushed to `<log_dir>/0.monitor.csv` in Monitor's format."""

    def __init__(self, num_envs, device, log_dir=None, env_id="", capacity=1 << 20):
        self.dev, self.log_dir, self.env_id, self.cap = device, log_dir, env_id, capacity
        self.ep_ret = torch.zeros(num_envs, device=device)
        self.ep_len = torch.zeros(num_envs, device=device)
        self.ring = torch.zeros(capacity + 1, 3, device=device)      # r, l, t; row `capacity` is a dump slot
        self.n_dev = torch.zeros((), dtype=torch.long, device=device)   # episodes recorded so far (device counter)
        self.flushed = 0
        self.t_start = time.time()
        self._file = None
        if log_dir is not None:
            os.makedirs(log_dir, exist_ok=True)
            self._file = os.path.join(log_dir, "0.monitor.csv")
            with open(self._file, "w") as f:
                f.write("#%s\n" % json.dumps({"t_start": self.t_start, "env_id": env_id}))
                f.write("r,l,t\n")

    @property
    def n(self):
        return int(self.n_dev.item())

    def update(self, raw_reward, done, count=None):
        """count: bool [E] rows that are real transitions (lazy auto-reset: ~resetting).  No host synchronisation:
        finished envs scatter their row to ring[n + rank among this call's finished envs], the others to the dump slot."""
        c = torch.ones_like(raw_reward) if count is None else count.float()
        self.ep_ret += raw_reward * c
        self.ep_len += c
        fin = done != 0
        rank = torch.cumsum(fin.long(), 0) - 1
        pos = torch.where(fin, (self.n_dev + rank) % self.cap, torch.full_like(rank, self.cap))
        t = torch.full_like(self.ep_ret, time.time() - self.t_start)
        self.ring[pos] = torch.stack([self.ep_ret, self.ep_len, t], dim=1)
        self.n_dev += fin.sum()
        self.ep_ret = torch.where(fin, torch.zeros_like(self.ep_ret), self.ep_ret)
        self.ep_len = torch.where(fin, torch.zeros_like(self.ep_len), self.ep_len)

    def last(self, window=100):
        """(r, l) of the last `window` finished episodes, oldest first"""
        n = self.n
        k = min(window, n, self.cap)
        idx = (torch.arange(n - k, n, device=self.dev)) % self.cap
        return self.ring[idx, 0], self.ring[idx, 1]

    def mean_reward(self, window=100):
        r, _ = self.last(window)
        return float(r.mean()) if r.numel() else None

    def flush(self):
        """append the episodes recorded since the last flush to the CSV"""
        n = self.n
        if self._file is None or n == self.flushed:
            return
        lo = max(self.flushed, n - self.cap)
        idx = (torch.arange(lo, n, device=self.dev)) % self.cap
        rows = self.ring[idx].cpu().tolist()
        with open(self._file, "a") as f:
            for r, l, t in rows:
                f.write("%.6f,%d,%.6f\n" % (r, int(l), t))
        self.flushed = n


class SaveOnBestTrainingRewardCallback:
    """benchmark/train.py:16-47: every `check_freq` calls of env.step, the mean return of the last 100 finished episodes
    is compared with the best so far and the model is saved to `<log_dir>/best_model.safetensors` when it improved
    (policy weights + the VecNormalize statistics that belong to them)."""

    def __init__(self, check_freq, log_dir, monitor, verbose=1):
        self.check_freq, self.log_dir, self.monitor, self.verbose = int(check_freq), log_dir, monitor, verbose
        self.save_path = os.path.join(log_dir, "best_model.safetensors")
        self.best_mean_reward = -float("inf")
        self.n_calls = 0
        self.saves = 0
        os.makedirs(log_dir, exist_ok=True)

    def on_step(self, model, venv, num_timesteps):
        self.n_calls += 1
        if self.n_calls % self.check_freq:
            return True
        self.monitor.flush()
        # the reference's Monitor sees sequential episodes of 4 envs and averages the last 100 (benchmark/train.py:30-33);
        # with thousands of envs finishing in the same call the last 100 entries are one corner of one batch, so the window
        # is widened to at least one episode per env
        mean_reward = self.monitor.mean_reward(max(100, int(self.monitor.ep_ret.numel())))
        if mean_reward is None:
            return True
        if self.verbose > 0:
            print("Num timesteps: {}".format(num_timesteps))
            print("Best mean reward: {:.2f} - Last mean reward per episode: {:.2f}".format(self.best_mean_reward, mean_reward))
        if mean_reward > self.best_mean_reward:
            self.best_mean_reward = mean_reward
            if self.verbose > 0:
                print("Saving new best model to {}".format(self.save_path))
            save_model(self.save_path, model, venv)
            self.saves += 1
        return True


def save_model(path, model, venv):
    from safetensors.torch import save_file
    sd = {"policy." + k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    sd.update({"vecnormalize." + k: v.detach().cpu().contiguous() for k, v in venv.state_dict().items()})
    save_file(sd, path)


def load_model(path, model, venv):
    from safetensors.torch import load_file
    sd = load_file(path)
    model.load_state_dict({k[len("policy."):]: v for k, v in sd.items() if k.startswith("policy.")})
    venv.load_state_dict({k[len("vecnormalize."):]: v for k, v in sd.items() if k.startswith("vecnormalize.")})


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
          quiet=False, auto_reset=True, log_dir=None, check_freq=1000, env=None):
    """auto_reset="lazy" (PickAndPlace): transitions flagged info["resetting"] carry no reward and no gradient and cut the
    return like an episode end - the env spends them on its reset ticks (include/xarm_hip.h XARM_AUTO_RESET_LAZY).
    log_dir: Monitor CSV + best-model checkpoints every `check_freq` env.step calls + the final VecNormalize statistics
    (benchmark/train.py:74,99,107-108).  env: an already built VecEnv (tests)."""
    torch.manual_seed(seed)
    if env is None:
        import gym_xarm_amd
        env = gym_xarm_amd.make(env_id, num_envs=num_envs, seed=seed, config=config, auto_reset=auto_reset)
    num_envs = env.num_envs
    venv = VecNormalize(env, gamma=gamma)
    dev = env.device
    monitor = EpisodeMonitor(num_envs, dev, log_dir, env_id)
    callback = SaveOnBestTrainingRewardCallback(check_freq, log_dir, monitor, verbose=0 if quiet else 1) if log_dir else None
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
            monitor.update(raw, done, ~info["resetting"] if "resetting" in info else None)
            if callback is not None:
                callback.on_step(model, venv, (it - 1) * n_steps * num_envs + (len(obs_buf) + 1) * num_envs)
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
            if dev.type == "cuda":
                torch.cuda.synchronize()
            rec = {"update": it, "env_steps": it * n_steps * num_envs, "mean_raw_reward": (raw_sum / (log_every * n_steps)).item(),
                   "success_rate": (succ_sum / done_sum.clamp(min=1)).item(), "episodes": int(done_sum.item()),
                   "env_steps_per_sec": it * n_steps * num_envs / (time.perf_counter() - t0)}
            hist.append(rec)
            if not quiet:
                print(json.dumps(rec), flush=True)
            succ_sum.zero_(); done_sum.zero_(); raw_sum.zero_()
    monitor.flush()
    if log_dir:
        venv.save(os.path.join(log_dir, "vec_normalize.safetensors"))       # benchmark/train.py:107-108
    venv.monitor, venv.callback = monitor, callback
    if hasattr(torch.cuda, "synchronize") and dev.type == "cuda":
        torch.cuda.synchronize()
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
    ap.add_argument("--log-dir", default=None, help="Monitor CSV, best_model.safetensors (every --check-freq calls), vec_normalize.safetensors")
    ap.add_argument("--check-freq", type=int, default=1000)
    args = ap.parse_args()
    cfg = {"reward_type": args.reward_type, "GUI": False} if ("Reach" in args.env or "PickAndPlace" in args.env) else None
    if "Handover" in args.env and "NoGoal" not in args.env:
        cfg = {"reward_type": args.reward_type}
    model, venv, hist = train(args.env, args.num_envs, args.updates, config=cfg, auto_reset="lazy" if args.lazy_reset else True,
                              log_dir=args.log_dir, check_freq=args.check_freq)
    if args.save:
        save_model(args.save, model, venv)


if __name__ == "__main__":
    main()
