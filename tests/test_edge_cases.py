"""Edge cases at the boundary on the GPU: empty / partial reset masks, zero-length reward batches, invalid arguments,
env counts that are not multiples of the wavefront, and independence of an env from the batch it runs in."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENVS = [("XarmPDPickAndPlace-v0", 4), ("XarmReach-v0", 4), ("XarmPDHandover-v0", 8), ("XarmPDStackTower-v0", 8)]


@pytest.mark.parametrize("env_id,A", ENVS)
def test_masks_empty_batches_and_bad_arguments(env_id, A):
    import torch
    import gym_xarm_amd
    from gym_xarm_amd._native import XarmNativeError
    E = 37                                                       # not a multiple of 32 / 64
    env = gym_xarm_amd.make(env_id, num_envs=E, seed=2, auto_reset=False)
    env.reset()
    before = env.get_state().clone()
    env.reset(mask=torch.zeros(E, dtype=torch.uint8))            # empty mask: nothing changes
    assert torch.equal(env.get_state(), before)
    m = torch.zeros(E, dtype=torch.uint8)
    m[[0, 5, 36]] = 1
    for _ in range(2):
        env.step(torch.zeros(E, A))
    mid = env.get_state().clone()
    env.reset(mask=m)                                            # ragged mask: only those rows change
    after = env.get_state()
    changed = (after != mid).any(dim=1).cpu().numpy()
    assert changed[[0, 5, 36]].all() and not changed[np.setdiff1d(np.arange(E), [0, 5, 36])].any()
    g = env.goal_dim
    assert env.compute_reward(torch.zeros(0, g), torch.zeros(0, g)).shape == (0,)     # zero-length batch
    with pytest.raises(AssertionError):
        env.step(torch.zeros(E, A + 1))                          # the reference's `assert action.shape == (4,)`
    with pytest.raises(ValueError):
        env.reset(mask=torch.zeros(E + 1, dtype=torch.uint8))
    with pytest.raises(XarmNativeError):
        gym_xarm_amd.make(env_id, num_envs=0)
    env.close()


@pytest.mark.parametrize("env_id,A", ENVS)
def test_an_env_does_not_depend_on_its_batch(env_id, A):
    """every env of a small batch equals, bit for bit, the same global env id in a larger batch - whatever the grid
    size (ragged last workgroup, the XCD-contiguous block remap of the cooperative kernels with nwg % 8 != 0)"""
    import torch
    import gym_xarm_amd
    ref = None
    for E in (1000, 1, 33, 37 * 4 + 3, 999):
        env = gym_xarm_amd.make(env_id, num_envs=E, seed=5)
        env.reset()
        gen = torch.Generator(device=env.device)
        gen.manual_seed(1)
        for _ in range(3):
            a = torch.rand(1000, A, device=env.device, generator=gen)[:E] * 2 - 1
            obs, rew, done, info = env.step(a)
        got = torch.cat([obs["observation"], obs["achieved_goal"], obs["desired_goal"], rew[:, None]], dim=1).clone()
        ref = got if ref is None else ref
        assert torch.equal(got, ref[:E]), "envs differ at batch size %d" % E
        env.close()


def test_fast_pipeline_at_ragged_sizes():
    """k_step_fast + the hand-off (what large PickAndPlace batches step on) at env counts that leave a ragged last
    workgroup in the fast kernel and a ragged last row group in the hand-off kernel, with auto-reset on: every env equals,
    bit for bit, the same global env id in a larger batch of the same pipeline - whichever wavefront its hand-off shares"""
    import torch
    import gym_xarm_amd
    ref = None
    for E in (1000, 2, 63, 65, 37 * 4 + 3, 999):      # (a 1-env handle would be within step_coop_limit = 1: cooperative step)
        env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=5, step_coop_limit=1, reset_coop_limit=0)
        assert env.kernel_limits()[1] == 1
        env.reset()
        env.set_episode_steps(torch.arange(E, device=env.device) % 50)        # time-limit resets inside the step calls
        gen = torch.Generator(device=env.device)
        gen.manual_seed(1)
        for _ in range(4):
            a = torch.rand(1000, 4, device=env.device, generator=gen)[:E] * 2 - 1
            obs, rew, done, info = env.step(a)
        got = torch.cat([obs["observation"], obs["achieved_goal"], obs["desired_goal"], rew[:, None], env.get_state()], dim=1).clone()
        ref = got if ref is None else ref
        assert torch.isfinite(got).all() and torch.equal(got, ref[:E]), "envs differ at batch size %d" % E
        env.close()
