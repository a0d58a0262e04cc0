#!/usr/bin/env python3
"""Side-by-side harness (SURVEY.md 8c 'optional live oracle'): when PyBullet is importable, run the
build's own primitives-only URDF (tools/emit_primitives_urdf.py) through PyBullet with the reference's call sequence
(xarm_pick_and_place.py:53-103 scene, :199-218 _set_action, :107-119 step) next to the CPU oracle on the same spawn,
goal and action sequence, print the trajectory differences (EEF, object, reward first; joints second - a 7-dof IK
with a different null-space choice moves the elbow, not the hand) and time PyBullet on this machine's cores.

PyBullet is not installed in the build image or on the GPU boxes of this project, so every run so far prints
    reference: unavailable (pybullet not importable)
and exits 0; nothing is installed or fetched.  This is the one route by which the physics constants restated from
memory (motor gains, ERPs, default inertia of inertial-less links, damping) can ever be pinned.
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(steps=50, seed=0):
    try:
        import pybullet as p
    except Exception as e:
        print("reference: unavailable (pybullet not importable: %s)" % type(e).__name__)
        return 0
    import json
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from emit_primitives_urdf import emit
    js = json.load(open(os.path.join(ROOT, "gym_xarm_amd", "model", "xarm7_pd.json")))
    c = js["pick_and_place"]
    urdf = os.path.join(tempfile.mkdtemp(), "xarm7_pd_primitives.urdf")
    open(urdf, "w").write(emit(js))
    p.connect(p.DIRECT)
    p.setTimeStep(c["time_step"])
    p.setPhysicsEngineParameter(numSubSteps=c["n_substeps"])
    t = js["table"]
    tc = p.createCollisionShape(p.GEOM_BOX, halfExtents=[t["half_x"], t["half_y"], 0.025])
    p.createMultiBody(0, tc, basePosition=[0, 0, t["top_z"] - 0.025])
    box = p.createMultiBody(c["obj_mass"], p.createCollisionShape(p.GEOM_BOX, halfExtents=c["obj_half"]))
    arm = p.loadURDF(urdf, [0, 0, 0], [0, 0, 0, 1], useFixedBase=True)
    names = {p.getJointInfo(arm, i)[12].decode(): i for i in range(p.getNumJoints(arm))}
    eef, hand, f1, f2 = names["link_eef"], names["panda_hand"], names["panda_leftfinger"], names["panda_rightfinger"]
    g = p.createConstraint(arm, f1, arm, f2, jointType=p.JOINT_GEAR, jointAxis=[1, 0, 0], parentFramePosition=[0, 0, 0], childFramePosition=[0, 0, 0])
    p.changeConstraint(g, gearRatio=-1, erp=0.1, maxForce=50)
    ora = O.OraclePnP(1, seed=seed)
    ora.reset()
    st = ora.get_state()[0]
    arm_joints = [names["link%d" % k] for k in range(1, 8)]
    for k, j in enumerate(arm_joints + [f1, f2]):
        p.resetJointState(arm, j, st[k], st[9 + k])
    p.resetBasePositionAndOrientation(box, st[18:21], st[21:25])
    rng = np.random.default_rng(seed)
    err = []
    t0 = time.perf_counter()
    for s in range(steps):
        a = rng.uniform(-1, 1, 4)
        cur = np.array(p.getLinkState(arm, eef)[0])
        tgt = np.clip(cur + a[:3] * c["max_vel"] * c["action_dt"], c["pos_low"], c["pos_high"])
        fg = float(np.clip(p.getJointState(arm, f1)[0] + a[3] * c["action_dt"] * c["max_gripper_vel"], c["gripper_low"], c["gripper_high"]))
        q = p.calculateInverseKinematics(arm, eef, tgt, [1, 0, 0, 0], maxNumIterations=c["n_substeps"])
        for k, j in enumerate(arm_joints):
            p.setJointMotorControl2(arm, j, p.POSITION_CONTROL, q[k])
        for j in (f1, f2):
            p.setJointMotorControl2(arm, j, p.POSITION_CONTROL, fg, force=c["finger_motor_force"])
        grasp = len(p.getContactPoints(arm, box, f1)) != 0 and len(p.getContactPoints(arm, box, f2)) != 0
        for j in (f1, f2):
            p.changeDynamics(arm, j, lateralFriction=100 if grasp else 1)
        p.setGravity(0, 0, -js["solver"]["gravity"])
        p.stepSimulation()
        ora.step(a[None])
        o = ora.get_state()[0]
        pe = np.array(p.getLinkState(arm, eef)[0])
        pb = np.array(p.getBasePositionAndOrientation(box)[0])
        qj = np.array([p.getJointState(arm, j)[0] for j in arm_joints])
        err.append((np.abs(pe - O.fk(o[:9])[0][7]).max(), np.abs(pb - o[18:21]).max(), np.abs(qj - o[:7]).max()))
    dt = time.perf_counter() - t0
    err = np.array(err)
    print("reference: pybullet %s, %d steps: max |EEF diff| %.3e m, max |object diff| %.3e m, max |joint diff| %.3e rad; %.0f env steps/s on one core"
          % (getattr(p, "__version__", "?"), steps, err[:, 0].max(), err[:, 1].max(), err[:, 2].max(), steps / dt))
    return 0


if __name__ == "__main__":
    sys.exit(main())
