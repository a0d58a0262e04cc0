"""XarmReach-v0 (reference xarm_reach.py): oracle pinned by the reference's own NumPy code, the kernel
core (host build) against the oracle, and - on the GPU - the HIP path through the C ABI against both.

Float tolerance: the six gripper joints carry 2e-5 .. 4e-4 kg m^2 behind motors that ask for up to
800 rad/s (max_gripper_vel 20, :29,137) and slam into their [0, 0.85] limits; their velocities are not
observable and chaotic, so parity is asserted on what the env exposes - the 8 observations, the arm
joints, reward / done / success / future_length - with per-field tolerances scaled by the oracle's
sensitivity (1e-6 input perturbation), as for PickAndPlace (oracle/parity.py)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# obs: hand COM pos 3, hand COM vel 3, driver q, driver qd
OBS_ATOL = np.array([1e-4] * 3 + [2e-2] * 3 + [5e-3, 60.0])
K_SENS = 300.0


@pytest.fixture(scope="module")
def gref():
    return np.load(os.path.join(GOLDEN, "reach_reward_reference.npz"))


@pytest.fixture(scope="module")
def groll():
    return np.load(os.path.join(GOLDEN, "reach_oracle_rollout.npz"))


def check_obs(obs, ref, sens, what):
    err = np.abs(np.asarray(obs, dtype=np.float64) - ref)
    ok = (err <= OBS_ATOL + 2e-4 * np.abs(ref) + K_SENS * sens[:, None]).all(axis=1) | (sens > 0.05)
    assert ok.all(), "%s: envs %s err %s sens %s" % (what, np.where(~ok)[0][:4], err[~ok][:4], sens[~ok][:4])
    assert (sens > 0.05).mean() <= 0.2


def test_rewards_match_reference_golden(oracle, gref):
    env = oracle.OracleReach(1)
    out = env.compute_reward(gref["achieved_goal"], gref["goal"], "sparse")
    assert np.array_equal(out.astype(np.float32), gref["reward_sparse"].astype(np.float32))
    assert np.array_equal(out.astype(np.float32), gref["is_success"])
    np.testing.assert_allclose(env.compute_reward(gref["achieved_goal"], gref["goal"], "dense"), gref["reward_dense"], atol=1e-15)
    with pytest.raises(ValueError):
        env.compute_reward(gref["achieved_goal"], gref["goal"], "dense_diff")


def test_dense_diff_done_future_length_through_step(oracle, gref):
    """the stateful reward and the step bookkeeping, driven through xo_reach_step"""
    env = oracle.OracleReach(2, seed=9, reward_type="dense_diff")
    obs, ag, dg = env.reset()
    d_old = np.linalg.norm(ag - dg, axis=1)
    np.testing.assert_allclose(env.state[:, 42], d_old, atol=1e-15)          # :100
    rng = np.random.default_rng(1)
    for k, s in enumerate(gref["steps"]):
        obs, ag, dg, rew, done, succ, fut = env.step(rng.uniform(-1, 1, (2, 4)))
        d = np.linalg.norm(ag - dg, axis=1)
        np.testing.assert_allclose(rew, d_old - d, atol=1e-15)               # :113-116
        d_old = d
        assert (done == gref["done"][k]).all() and (fut == gref["future_length"][k]).all()
        assert np.array_equal(succ, (d < 0.05).astype(np.uint8))
    # the reference's own dense_diff sequence, replayed on the same distances
    d0, rr = float(gref["diff_d_old0"]), []
    for p in gref["diff_traj"]:
        d = np.linalg.norm(p - gref["diff_goal"])
        rr.append(d0 - d)
        d0 = d
    np.testing.assert_allclose(rr, gref["reward_dense_diff"], atol=1e-15)


def test_oracle_reach_behaviour(oracle):
    env = oracle.OracleReach(4, seed=1)
    obs, ag, dg = env.reset()
    js = oracle.load_model_json(oracle.REACH_JSON)["reach"]
    assert (dg >= js["goal_low"]).all() and (dg < js["goal_high"]).all()
    for k in range(25):
        a = np.zeros((4, 4))
        a[:, :3] = np.clip((dg - ag) * 12, -1, 1)
        a[:, 3] = 0.3
        obs, ag, dg, rew, done, succ, fut = env.step(a)
    assert (np.linalg.norm(ag - dg, axis=1) < 1e-3).all() and (rew == 1).all() and done.all() and (fut == 0).all()
    assert (obs[:, 6] > 0.84).all() and (obs[:, 6] < 0.86).all()       # driver joint parked at its upper limit 0.85
    big, lo, hi = oracle.OracleReach(8, seed=2), oracle.OracleReach(4, seed=2), oracle.OracleReach(4, seed=2, env_id_offset=4)
    assert np.array_equal(big.state[:4], lo.state) and np.array_equal(big.state[4:], hi.state)


def test_reach_model_matches_urdf(oracle):
    import xml.etree.ElementTree as ET
    urdf = "/root/reference/gym_xarm/envs/urdf/xarm7.urdf"
    if not os.path.exists(urdf):
        pytest.skip("reference not present")
    js = oracle.load_model_json(oracle.REACH_JSON)
    root = ET.parse(urdf).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.find("child").get("link"): j for j in root.findall("joint")}
    names = [l["name"] for l in js["links"]]
    for l in js["links"]:
        j = joints[l["name"]]
        assert j.get("type") == l["joint"]
        assert j.find("parent").get("link") == ("link_base" if l["parent"] < 0 else names[l["parent"]])
        np.testing.assert_allclose([float(x) for x in j.find("origin").get("xyz").split()], l["origin_xyz"], atol=0)
        if l["joint"] != "fixed":
            np.testing.assert_allclose([float(x) for x in j.find("axis").get("xyz").split()], l["axis"], atol=0)
            assert float(j.find("limit").get("lower")) == l["lower"] and float(j.find("limit").get("upper")) == l["upper"]
        inert = links[l["name"]].find("inertial")
        if inert is None:
            assert l["mass"] == 1.0
            continue
        assert float(inert.find("mass").get("value")) == l["mass"]
        np.testing.assert_allclose([float(x) for x in inert.find("origin").get("xyz").split()], l["com"], atol=0)
        I = inert.find("inertia")
        np.testing.assert_allclose([float(I.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")], l["inertia"], atol=0)
    assert names.index("left_outer_knuckle") + 1 == 10 and len(names) + 1 == 17    # gripper_driver_index, num_joints (:21-22)


@pytest.mark.parametrize("coop", [False, True], ids=["lane", "coop"])
def test_hostcore_f64_equals_oracle(hostcore, groll, coop):
    """both kernel families' cores (one env per lane: xarm_reach_core.h; one env per 16-lane row:
    xarm_reach_coop_core.h) instantiated in float64 on the host reproduce the oracle"""
    g = groll
    st = hostcore.reach_init(32, f32=0, seed=3)
    np.testing.assert_allclose(st, g["init_state"], atol=1e-15)
    st, obs, ag, dg = hostcore.reach_reset(st, f32=0, seed=3, coop=coop)
    np.testing.assert_allclose(st, g["states"][0], atol=1e-10)
    np.testing.assert_allclose(obs, g["reset_obs"], atol=1e-10)
    for t in range(g["actions"].shape[0]):
        st, obs, ag, dg, rew, done, succ, fut = hostcore.reach_step(g["states"][t], g["actions"][t], f32=0, seed=3, coop=coop)
        ok = g["sens"][t] < 1e-3
        np.testing.assert_allclose(st[ok, :7], g["states"][t + 1][ok, :7], atol=1e-8)
        np.testing.assert_allclose(obs[ok][:, :7], g["obs"][t][ok][:, :7], atol=1e-6)
        assert np.array_equal(rew[ok], g["rew"][t][ok]) and np.array_equal(done, g["done"][t]) and np.array_equal(fut, g["fut"][t])
        assert ok.mean() > 0.7


@pytest.mark.parametrize("coop", [False, True], ids=["lane", "coop"])
def test_hostcore_f32_within_tolerance(hostcore, groll, coop):
    g = groll
    for t in range(g["actions"].shape[0]):
        st, obs, ag, dg, rew, done, succ, fut = hostcore.reach_step(g["states"][t], g["actions"][t], f32=1, seed=3, coop=coop)
        check_obs(obs, g["obs"][t], g["sens"][t], "f32 host t=%d" % t)
        ok = g["sens"][t] < 1e-3
        np.testing.assert_allclose(st[ok, :7], g["states"][t + 1][ok, :7], atol=1e-4)
        assert np.array_equal(done, g["done"][t]) and np.array_equal(fut, g["fut"][t])


# ------------------------------------------------------------------------------------------- GPU
# kernel families: "lane" = k_reach_step / k_reach_reset (one env per lane), "coop" = k_reach_step_coop /
# k_reach_reset_coop (one env per DPP row); the limits force one or the other whatever the batch size
FAMILY = {"lane": dict(reset_coop_limit=-1, step_coop_limit=-1), "coop": dict(reset_coop_limit=1 << 30, step_coop_limit=1 << 30)}


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["lane", "coop"])
def test_gpu_reach_replays_golden_rollout(groll, family):
    import torch
    import gym_xarm_amd as gx
    g = groll
    E = g["states"].shape[1]
    env = gx.make("XarmReach-v0", num_envs=E, seed=3, auto_reset=False, **FAMILY[family])
    np.testing.assert_allclose(env.get_state().cpu().numpy(), g["init_state"], atol=1e-6)
    obs = env.reset()
    np.testing.assert_allclose(obs["observation"].cpu().numpy()[:, :3], g["reset_obs"][:, :3], atol=1e-4)
    np.testing.assert_allclose(obs["desired_goal"].cpu().numpy(), g["states"][0][:, 39:42], atol=1e-6)
    for t in range(g["actions"].shape[0]):
        env.set_state(g["states"][t])
        obs, rew, done, info = env.step(torch.tensor(g["actions"][t], dtype=torch.float32))
        sens = g["sens"][t]
        check_obs(obs["observation"].cpu().numpy(), g["obs"][t], sens, "gpu t=%d" % t)
        st = env.get_state().cpu().numpy().astype(np.float64)
        ok = sens < 1e-3
        np.testing.assert_allclose(st[ok, :7], g["states"][t + 1][ok, :7], atol=1e-4)
        well = ok & (np.abs(np.linalg.norm(g["obs"][t][:, :3] - g["states"][t][:, 39:42], axis=1) - 0.05) > 1e-3)
        assert np.array_equal(rew.cpu().numpy()[well], g["rew"][t][well].astype(np.float32))
        assert np.array_equal(done.cpu().numpy(), g["done"][t])
        assert np.array_equal(info["future_length"].cpu().numpy(), g["fut"][t])
        assert np.array_equal(info["is_success"].cpu().numpy()[well], g["succ"][t][well])
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["lane", "coop", "default"])
def test_gpu_reach_4096_episode_and_registry(gref, family):
    """BASELINE config 2: XarmReach-v0, 4096 envs on one GPU; plus the reference's test.py pattern"""
    import torch
    import gym_xarm_amd as gx
    E = 4096
    env = gx.make("XarmReach-v0", num_envs=E, seed=0, config={"reward_type": "dense", "GUI": False}, **FAMILY.get(family, {}))
    obs = env.reset()
    dg = obs["desired_goal"].clone()
    for k in range(25):
        a = torch.zeros(E, 4, device="cuda")
        a[:, :3] = ((obs["desired_goal"] - obs["achieved_goal"]) * 12).clamp(-1, 1)
        obs, rew, done, info = env.step(a)
        assert torch.isfinite(obs["observation"]).all()
        if k < 24:
            assert not done.any() and (info["future_length"] == 24 - k).all()
            np.testing.assert_allclose(rew.cpu().numpy(), -(obs["achieved_goal"] - obs["desired_goal"]).norm(dim=1).cpu().numpy(), atol=1e-6)
    assert done.all() and info["TimeLimit.truncated"].all()
    term = info["terminal_observation"]
    assert ((term[:, :3] - dg).norm(dim=1) < 2e-3).all()              # the P-controller reached every goal
    assert (info["future_length"] == 25).all()                        # auto-reset: fresh episodes
    assert ((obs["desired_goal"] - dg).abs().max(dim=1).values > 0).float().mean() > 0.99   # new goals
    out = env.compute_reward(torch.tensor(gref["achieved_goal"], dtype=torch.float32), torch.tensor(gref["goal"], dtype=torch.float32))
    np.testing.assert_allclose(out.cpu().numpy(), gref["reward_dense"], atol=1e-6)
    env.close()
    one = gx.make("XarmReach-v0", config={"reward_type": "sparse", "GUI": False})
    ob = one.reset()
    for i in range(27):
        assert one.observation_space.contains(ob)
        a = one.action_space.sample()
        ob, r, d, info = one.step(a)
        assert ob["observation"].shape == (8,) and d == (i % 25 == 24) and info["future_length"] == 24 - i % 25
        if d:
            ob = one.reset()
    one.close()


@pytest.mark.gpu
def test_gpu_reach_kernel_families_agree_and_limits():
    """the two kernel families step the same 4 096 envs through a whole episode + auto-reset and stay together
    (observable part of the state; the gripper joint rates are chaotic); `kernel_limits` reports what
    the handle runs on: default 8 192 / 8 192, < 0 = never, the environment variables override"""
    import subprocess
    import sys
    import torch
    import gym_xarm_amd as gx
    E = 4096
    envs = {f: gx.make("XarmReach-v0", num_envs=E, seed=9, config={"reward_type": "dense", "GUI": False}, **FAMILY[f]) for f in ("lane", "coop")}
    assert envs["lane"].kernel_limits() == (0, 0) and envs["coop"].kernel_limits() == (1 << 30, 1 << 30)
    obs = {f: e.reset() for f, e in envs.items()}
    gen = torch.Generator(device="cuda").manual_seed(3)
    worst = worst_v = 0.0
    for t in range(30):
        a = torch.rand(E, 4, device="cuda", generator=gen) * 2 - 1
        out = {f: e.step(a) for f, e in envs.items()}
        (ol, rl, dl, il), (oc, rc, dc, ic) = out["lane"], out["coop"]
        assert torch.equal(dl, dc) and torch.equal(il["future_length"], ic["future_length"])
        assert torch.equal(ol["desired_goal"], oc["desired_goal"])                 # same RNG stream through the reset
        d = (ol["observation"][:, :3] - oc["observation"][:, :3]).abs().max().item()
        worst = max(worst, d)
        dv = (ol["observation"][:, 3:6] - oc["observation"][:, 3:6]).abs().max(dim=1).values
        worst_v = max(worst_v, torch.quantile(dv, 0.99).item())
        assert (rl - rc).abs().max().item() < 2e-3
    # free-running: the hand position stays together; the hand velocity feels the chaotic gripper joints
    print("lane vs coop Reach kernels over 30 steps: worst |d hand pos| = %.2e, 99 %% quantile of |d hand vel| <= %.2e" % (worst, worst_v))
    assert worst < 5e-3 and worst_v < 0.2     # measured on MI355X: 1.6e-3 m, 6.8e-2 m/s
    for e in envs.values():
        e.close()
    d = gx.make("XarmReach-v0", num_envs=8)
    assert d.kernel_limits() == (8192, 8192)
    d.close()
    code = ("import gym_xarm_amd as gx; e = gx.make('XarmReach-v0', num_envs=8); print(e.kernel_limits()); e.close()")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, XARM_RESET_COOP_LIMIT="0", XARM_STEP_COOP_LIMIT="123", PYTHONPATH=ROOT))
    assert out.returncode == 0 and "(0, 123)" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_gpu_a2c_learns_reach(tmp_path):
    """the reference's consumer of the env boundary (benchmark/train.py: VecNormalize + A2C) on the batched env:
    a few hundred on-device A2C updates must improve the dense Reach reward markedly; the run leaves the artefacts of
    the reference's script (Monitor CSV :74, best model :16-47, VecNormalize statistics :107-108)"""
    import os
    from gym_xarm_amd.train import train
    model, venv, hist = train("XarmReach-v0", num_envs=2048, updates=300, config={"reward_type": "dense", "GUI": False},
                              log_every=50, quiet=True, log_dir=str(tmp_path), check_freq=250)
    first, last = hist[0]["mean_raw_reward"], hist[-1]["mean_raw_reward"]
    assert last > first + 0.03, (first, last, hist)          # mean -distance to the goal shrinks by > 3 cm
    assert hist[-1]["env_steps_per_sec"] > 2e5
    assert {"0.monitor.csv", "best_model.safetensors", "vec_normalize.safetensors"} <= set(os.listdir(str(tmp_path)))
    rows = open(os.path.join(str(tmp_path), "0.monitor.csv")).read().strip().split("\n")
    assert rows[1] == "r,l,t" and len(rows) - 2 == venv.monitor.n == 2048 * (1500 // 25)       # every 25-step episode logged
    r = np.array([float(x.split(",")[0]) for x in rows[2:]])
    l = np.array([int(x.split(",")[1]) for x in rows[2:]])
    assert (l == 25).all() and r[-2048:].mean() > r[:2048].mean()                                 # returns improve
    assert venv.callback.saves >= 2 and venv.callback.best_mean_reward > r[:2048].mean()
