"""Analytic known answers for the restated physics (PyBullet is absent, so these are the only external pins of the
dynamics): free fall, Coulomb sliding, resting contact, and the arm holding a pose against gravity."""
import numpy as np

G, DT_SUB, N_SUB = 9.8, 1.0 / 900.0, 15


def _box_env(oracle, p, v=(0, 0, 0), w=(0, 0, 0)):
    env = oracle.OraclePnP(1, seed=0)
    s = env.get_state()
    s[0, 18:21] = p
    s[0, 21:25] = [0, 0, 0, 1]
    s[0, 25:28] = v
    s[0, 28:31] = w
    s[0, 34:50] = 0
    env.set_state(s)
    return env


def test_free_fall_matches_the_discrete_closed_form(oracle):
    """semi-implicit Euler with Bullet's damping: v_{k+1} = (v_k - g dt) (1 - 0.04)^dt, z_{k+1} = z_k + dt v_{k+1}"""
    env = _box_env(oracle, [0.8, 0.0, 3.0])            # beside the table: nothing to hit for a while
    z, v = 3.0, 0.0
    damp = (1.0 - 0.04) ** DT_SUB
    for step in range(10):
        env.step(np.zeros((1, 4)))
        for _ in range(N_SUB):
            v = (v - G * DT_SUB) * damp
            z += DT_SUB * v
        st = env.get_state()[0]
        assert abs(st[20] - z) < 1e-12 and abs(st[27] - v) < 1e-12
    assert abs(z - (3.0 - 0.5 * G * (10 / 60) ** 2)) < 2e-3      # and that is the parabola up to O(dt) and damping


def test_sliding_box_obeys_coulomb_friction(oracle):
    """box on the table with 0.6 m/s along x: decelerates at mu g (mu = mu_object * mu_table = 0.5) and stops after
    v0^2 / (2 mu g); the PGS friction rows are box-clamped per direction, exact for motion along one tangent axis"""
    v0, mu = 0.6, 0.5
    env = _box_env(oracle, [0.1, 0.0, 0.04], v=[v0, 0, 0])        # far from the gripper (x = 0.1), flat on the table
    xs, vs = [], []
    for _ in range(12):
        env.step(np.zeros((1, 4)))
        st = env.get_state()[0]
        xs.append(st[18]); vs.append(st[25])
    vs = np.array(vs)
    t = (np.arange(12) + 1) / 60.0
    moving = vs > 0.05
    assert moving.sum() >= 5
    decel = -(np.diff(vs[moving]) * 60.0)
    assert np.all(np.abs(decel - mu * G) < 0.25), decel            # 4.9 m/s^2 (+ the small 4 %/s damping)
    assert abs(vs[-1]) < 1e-3                                      # at rest after 0.122 s
    assert abs((xs[-1] - 0.1) - v0 ** 2 / (2 * mu * G)) < 3e-3    # 36.7 mm
    st = env.get_state()[0]
    assert abs(st[20] - 0.04) < 2e-4 and np.abs(st[28:31]).max() < 1e-2   # did not tip or sink


def test_resting_box_stays_put_and_carries_its_weight(oracle):
    env = _box_env(oracle, [0.1, 0.1, 0.04])
    for _ in range(5):
        env.step(np.zeros((1, 4)))
    st = env.get_state()[0]
    assert np.abs(st[18:21] - [0.1, 0.1, 0.04]).max() < 1e-4 and np.abs(st[25:31]).max() < 1e-3
    # the four corner impulses of the last substep add up to m g dt (0.5 kg box)
    assert abs(st[34:42].sum() - 0.5 * G * DT_SUB) < 2e-5


def test_arm_holds_its_pose_against_gravity(oracle):
    """zero action: the Cartesian target stays at the current hand position, the velocity-level PD motors (max impulse
    1e5 / 60 per row) hold 4.6 kg of arm against gravity with sub-millimetre sag"""
    env = oracle.OraclePnP(1, seed=0)
    env.reset()
    for _ in range(20):            # the start pose (z = 0.12) is below the workspace floor (z = 0.15): the clip lifts the hand first
        env.step(np.zeros((1, 4)))
    hand0 = oracle.fk(env.get_state()[0, :9])[0][7].copy()
    assert abs(hand0[2] - 0.15) < 2e-3
    for _ in range(20):
        env.step(np.zeros((1, 4)))
    st = env.get_state()[0]
    assert np.abs(oracle.fk(st[:9])[0][7] - hand0).max() < 1e-3
    assert np.abs(st[9:16]).max() < 1e-2


# ------------------------------------------------------------------------------------------------ the HIP path itself
import pytest  # noqa: E402


@pytest.mark.gpu
def test_hip_path_free_fall_and_coulomb_sliding():
    """the same closed forms on the device (float32): not only 'equal to the oracle' but equal to the analytic answers"""
    import torch
    import gym_xarm_amd
    env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=64, seed=0, auto_reset=False)
    env.reset()
    s = env.get_state()
    s[:, 34:50] = 0
    s[:, 21:25] = torch.tensor([0., 0., 0., 1.], device=env.device)
    s[:, 28:31] = 0
    s[:32, 18:21] = torch.tensor([0.8, 0.0, 3.0], device=env.device); s[:32, 25:28] = 0              # free fall beside the table
    s[32:, 18:21] = torch.tensor([0.1, 0.0, 0.04], device=env.device)                                 # sliding on the table
    s[32:, 25:28] = torch.tensor([0.6, 0.0, 0.0], device=env.device)
    env.set_state(s)
    z, v, damp = 3.0, 0.0, (1.0 - 0.04) ** DT_SUB
    for _ in range(12):
        env.step(torch.zeros(64, 4))
        for _ in range(N_SUB):
            v = (v - G * DT_SUB) * damp
            z += DT_SUB * v
    st = env.get_state().cpu().numpy()
    assert np.abs(st[:32, 20] - z).max() < 2e-5 and np.abs(st[:32, 27] - v).max() < 2e-5
    assert np.abs(st[32:, 25]).max() < 1e-3                                                            # stopped
    assert np.abs((st[32:, 18] - 0.1) - 0.6 ** 2 / (2 * 0.5 * G)).max() < 3e-3                         # after 36.7 mm
    assert np.abs(st[32:, 20] - 0.04).max() < 3e-4
    env.close()
