"""Stand-in VecEnv for the CPU rehearsal of bench.py's rank plumbing (tests/test_distributed.py): the call surface
bench.py uses, arithmetic that depends on the global env id so that shard offsets are visible in the output."""
import json
import os

import torch


class StubEnv:
    max_episode_steps = 10

    def __init__(self, env_id, num_envs, seed, env_id_offset, device, config=None, auto_reset=True):
        self.num_envs, self.offset, self.device, self.auto_reset = int(num_envs), int(env_id_offset), device, auto_reset
        self.gid = torch.arange(self.num_envs) + self.offset
        self.steps = torch.zeros(self.num_envs, dtype=torch.long)
        self.n_step = 0
        log = os.environ.get("XARM_BENCH_STUB_LOG")
        if log:
            with open(log + ".%d" % int(os.environ.get("RANK", 0)), "a") as f:
                f.write(json.dumps({"env_id": env_id, "num_envs": self.num_envs, "env_id_offset": self.offset, "auto_reset": str(auto_reset)}) + "\n")

    def reset(self):
        self.steps.zero_()
        return {}

    def set_episode_steps(self, s):
        self.steps = torch.as_tensor(s).long().clone()

    def step(self, a):
        assert a.shape[0] == self.num_envs
        self.n_step += 1
        self.steps += 1
        done = (self.steps >= self.max_episode_steps).to(torch.uint8)
        self.steps = torch.where(done != 0, torch.zeros_like(self.steps), self.steps)
        info = {"is_success": done, "resetting": torch.zeros(self.num_envs, dtype=torch.bool)}
        return {}, torch.zeros(self.num_envs), done, info

    def timing_enable(self, on=True):
        self._t0 = self.n_step

    def timing_read(self):
        return 0.5 * (self.n_step - self._t0), self.n_step - self._t0

    def timing_read_reset(self):
        return 0.25 * (self.n_step - self._t0), self.n_step - self._t0

    def pipeline_info(self):
        return dict(fast_pipeline=False, reset_overlap=False, eject_coop_cap=0, solver_iterations=50)

    def library(self):
        return "stand-in env (tests/bench_stub.py)", None

    def close(self):
        pass


def make(env_id, num_envs, seed=0, env_id_offset=0, device=None, config=None, **kw):
    return StubEnv(env_id, num_envs, seed, env_id_offset, device, config, **kw)
