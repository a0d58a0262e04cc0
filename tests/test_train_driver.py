"""The training-side counterparts of benchmark/train.py that are not the A2C update itself: Monitor-style episode log
(:74), best-model-on-mean-reward checkpoint (:16-47) and VecNormalize save / load (:107-108, display.py:22).  CPU tests on
a scripted stand-in VecEnv (torch CPU tensors, same call surface as gym_xarm_amd's VecEnv); the GPU learning test in
tests/test_reach.py runs the same code on the real env."""
import os

import numpy as np
import pytest
import torch

from gym_xarm_amd.train import EpisodeMonitor, SaveOnBestTrainingRewardCallback, VecNormalize, ActorCritic, save_model, load_model, train


class ScriptedEnv:
    """E envs, episode length 3 + (e % 4), reward = 0.1 * (e + 1) * phase per step; auto-reset semantics"""

    def __init__(self, E, lazy=False):
        self.num_envs, self.obs_dim, self.goal_dim, self.act_dim = E, 5, 3, 2
        self.device = torch.device("cpu")
        self.t = torch.zeros(E)
        self.length = 3 + torch.arange(E) % 4
        self.lazy = lazy
        self.calls = 0

    def _obs(self):
        o = torch.stack([self.t + k for k in range(5)], dim=1)
        return {"observation": o, "achieved_goal": o[:, :3] * 0.5, "desired_goal": torch.ones(self.num_envs, 3)}

    def reset(self):
        self.t.zero_()
        return self._obs()

    def step(self, a):
        self.calls += 1
        self.t += 1
        rew = 0.1 * (torch.arange(self.num_envs) + 1.0) * self.t
        done = (self.t >= self.length).to(torch.uint8)
        self.t = torch.where(done != 0, torch.zeros_like(self.t), self.t)
        info = {"is_success": done.clone(), "TimeLimit.truncated": torch.zeros_like(done, dtype=torch.bool)}
        if self.lazy:
            info["resetting"] = (torch.arange(self.num_envs) % 2 == 0) & (self.calls % 3 == 0)
        return self._obs(), rew, done, info

    def close(self):
        pass


def test_monitor_rows_match_episode_returns(tmp_path):
    E = 6
    env = ScriptedEnv(E)
    mon = EpisodeMonitor(E, env.device, str(tmp_path), "Scripted-v0", capacity=16)
    env.reset()
    expect = []
    ret, ln = np.zeros(E), np.zeros(E)
    for k in range(9):
        obs, rew, done, info = env.step(None)
        mon.update(rew, done)
        ret += rew.numpy(); ln += 1
        for e in np.nonzero(done.numpy())[0]:
            expect.append((ret[e], ln[e])); ret[e] = 0; ln[e] = 0
    assert mon.n == len(expect) > 10
    mon.flush()
    lines = open(os.path.join(str(tmp_path), "0.monitor.csv")).read().strip().split("\n")
    assert lines[0].startswith("#{") and '"env_id": "Scripted-v0"' in lines[0] and lines[1] == "r,l,t"      # SB3 Monitor layout
    rows = np.array([[float(x) for x in l.split(",")] for l in lines[2:]])
    assert rows.shape == (len(expect), 3)
    assert np.allclose(sorted(rows[:, 0]), sorted(r for r, _ in expect), atol=1e-5)
    assert sorted(rows[:, 1]) == sorted(l for _, l in expect) and (np.diff(rows[:, 2]) >= 0).all()
    r, l = mon.last(4)
    assert np.allclose(r.numpy(), rows[-4:, 0], atol=1e-5)
    # ring wrap-around keeps the newest `capacity` episodes, flush appends only new rows
    for k in range(20):
        obs, rew, done, info = env.step(None)
        mon.update(rew, done)
    mon.flush()
    n_lines = len(open(os.path.join(str(tmp_path), "0.monitor.csv")).read().strip().split("\n")) - 2
    assert n_lines <= mon.n and mon.mean_reward(100) is not None and mon.last(100)[0].numel() == 16


def test_best_model_callback_saves_only_on_improvement(tmp_path):
    E = 4
    env = ScriptedEnv(E)
    venv = VecNormalize(env)
    mon = EpisodeMonitor(E, env.device, str(tmp_path), "Scripted-v0")
    model = ActorCritic(venv.dim, env.act_dim)
    cb = SaveOnBestTrainingRewardCallback(check_freq=5, log_dir=str(tmp_path), monitor=mon, verbose=0)
    venv.reset()
    scale = [1.0] * 10 + [3.0] * 10 + [0.5] * 10          # mean episode reward rises, then falls
    for k in range(30):
        obs, nrew, done, info, raw = venv.step(None)
        mon.update(raw * scale[k], done)
        cb.on_step(model, venv, (k + 1) * E)
    assert cb.n_calls == 30 and 1 <= cb.saves <= 4 and cb.best_mean_reward > 0
    assert os.path.exists(cb.save_path)
    saves_before = cb.saves
    for k in range(10):                                   # worse episodes: no new checkpoint
        obs, nrew, done, info, raw = venv.step(None)
        mon.update(raw * 0.01, done)
        cb.on_step(model, venv, 0)
    assert cb.saves == saves_before
    # the checkpoint restores the policy and the normalisation statistics that belong to it
    model2, venv2 = ActorCritic(venv.dim, env.act_dim), VecNormalize(ScriptedEnv(E))
    load_model(cb.save_path, model2, venv2)
    x = torch.randn(3, venv.dim)
    assert venv2.obs_rms.count > 1 and venv2.obs_rms.mean.abs().sum() > 0
    assert model2.pi[0].weight.shape == model.pi[0].weight.shape


def test_vecnormalize_save_load_roundtrip_and_lazy_mask(tmp_path):
    env = ScriptedEnv(8, lazy=True)
    venv = VecNormalize(env, clip_obs=7.0, gamma=0.9)
    venv.reset()
    seen = 8
    for k in range(12):
        _, _, _, info, _ = venv.step(None)
        seen += int((~info["resetting"]).sum())
    # rows flagged `resetting` (lazy auto-reset ticks) never enter the running statistics
    assert abs(venv.obs_rms.count - seen) < 1e-3
    path = str(tmp_path / "vec_normalize.safetensors")
    venv.save(path)
    v2 = VecNormalize.load(path, ScriptedEnv(8))
    assert torch.equal(v2.obs_rms.mean, venv.obs_rms.mean) and torch.equal(v2.obs_rms.var, venv.obs_rms.var)
    assert v2.obs_rms.count == venv.obs_rms.count and v2.ret_rms.count == venv.ret_rms.count
    assert torch.equal(v2.ret_rms.var, venv.ret_rms.var) and v2.clip_obs == 7.0 and abs(v2.gamma - 0.9) < 1e-7
    v2.training = False
    o = ScriptedEnv(8).reset()
    a, b = venv._flat(o), v2._flat(o)
    venv.training = False
    assert torch.equal(venv._norm(a), v2._norm(b))
    class Other(ScriptedEnv):
        def __init__(self):
            super().__init__(8); self.obs_dim = 9
    with pytest.raises(ValueError):
        VecNormalize.load(path, Other())


def test_train_loop_writes_monitor_checkpoint_and_stats(tmp_path):
    """the whole driver on the stand-in env (CPU): log_dir gets the three artefacts of benchmark/train.py"""
    env = ScriptedEnv(16)
    model, venv, hist = train(env_id="Scripted-v0", updates=12, n_steps=5, log_every=6, quiet=True, log_dir=str(tmp_path), check_freq=10, env=env)
    files = set(os.listdir(str(tmp_path)))
    assert {"0.monitor.csv", "best_model.safetensors", "vec_normalize.safetensors"} <= files
    assert venv.callback.n_calls == 60 and venv.monitor.n > 100 and len(hist) == 2
    rows = open(os.path.join(str(tmp_path), "0.monitor.csv")).read().strip().split("\n")
    assert len(rows) - 2 == venv.monitor.n


def test_vecnormalize_loads_statistics_files_without_the_later_keys():
    """files written before the return statistics / clip settings were saved hold obs_* only: load keeps the wrapper's own
    values for what the file lacks instead of raising KeyError"""
    env = ScriptedEnv(4)
    a, b = VecNormalize(env, clip_obs=7.0, gamma=0.9), VecNormalize(env, clip_obs=3.0, gamma=0.5)
    a.reset()
    for _ in range(5):
        a.step(torch.zeros(4, 2))
    sd = {k: v for k, v in a.state_dict().items() if k.startswith("obs_")}
    b.load_state_dict(sd)
    assert torch.equal(b.obs_rms.mean, a.obs_rms.mean) and b.clip_obs == 3.0 and b.gamma == 0.5
    b.load_state_dict(a.state_dict())
    assert b.clip_obs == 7.0 and abs(b.gamma - 0.9) < 1e-6 and torch.equal(b.ret_rms.var, a.ret_rms.var)      # gamma travels as float32


def test_best_model_window_covers_every_env():
    """with thousands of envs finishing in one call the reference's 'last 100 episodes' (benchmark/train.py:30-33) would be
    one corner of one batch: the callback averages over at least one episode per env"""
    E = 300
    mon = EpisodeMonitor(E, torch.device("cpu"))
    rew = torch.arange(E, dtype=torch.float32)
    mon.update(rew, torch.ones(E, dtype=torch.uint8))      # all E envs finish in the same call
    assert abs(mon.mean_reward(100) - float(rew[-100:].mean())) < 1e-4      # the biased corner
    assert abs(mon.mean_reward(max(100, E)) - float(rew.mean())) < 1e-4
