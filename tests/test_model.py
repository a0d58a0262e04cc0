"""The flat model table (gym_xarm_amd/model/xarm7_pd.json) against the reference URDF and env file
(only where /root/reference exists, i.e. in the build container), and the generated kernel header
against the table."""
import json
import os
import re
import subprocess
import sys
import xml.etree.ElementTree as ET

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
URDF = "/root/reference/gym_xarm/envs/urdf/xarm7_pd.urdf"
PNP = "/root/reference/gym_xarm/envs/xarm_pick_and_place.py"
JS = json.load(open(os.path.join(ROOT, "gym_xarm_amd", "model", "xarm7_pd.json")))


def test_generated_header_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_model_header.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(URDF), reason="reference not present on this machine")
def test_links_match_urdf():
    root = ET.parse(URDF).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.find("child").get("link"): j for j in root.findall("joint")}
    names = [l["name"] for l in JS["links"]]
    for i, l in enumerate(JS["links"]):
        j = joints[l["name"]]
        assert j.get("type") == l["joint"]
        parent = j.find("parent").get("link")
        assert parent == ("link_base" if l["parent"] < 0 else names[l["parent"]])
        org = j.find("origin")
        np.testing.assert_allclose([float(x) for x in org.get("xyz").split()], l["origin_xyz"], atol=0)
        np.testing.assert_allclose([float(x) for x in org.get("rpy").split()], l["origin_rpy"], atol=0)
        if l["joint"] != "fixed":
            np.testing.assert_allclose([float(x) for x in j.find("axis").get("xyz").split()], l["axis"], atol=0)
            lim = j.find("limit")
            assert float(lim.get("lower")) == l["lower"] and float(lim.get("upper")) == l["upper"]
            dyn = j.find("dynamics")
            assert (float(dyn.get("damping")) if dyn is not None else 0.0) == l["damping"]
        inert = links[l["name"]].find("inertial")
        if inert is None:
            assert l["name"] == "link_eef" and l["mass"] == 1.0 and l["inertia"] == [1.0, 0, 0, 1.0, 0, 1.0]
            continue
        assert float(inert.find("mass").get("value")) == l["mass"]
        np.testing.assert_allclose([float(x) for x in inert.find("origin").get("xyz").split()], l["com"], atol=0)
        I = inert.find("inertia")
        np.testing.assert_allclose([float(I.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")], l["inertia"], atol=0)
    # Bullet joint indices used by the reference (:31-36): eef 8, hand 9, fingers 10/11
    assert names.index("link_eef") + 1 == 8 and names.index("panda_hand") + 1 == 9
    assert names.index("panda_leftfinger") + 1 == 10 and names.index("panda_rightfinger") + 1 == 11


@pytest.mark.skipif(not os.path.exists(PNP), reason="reference not present on this machine")
def test_scene_constants_match_env_file():
    src = open(PNP).read()
    p = JS["pick_and_place"]

    def grab(pattern):
        m = re.search(pattern, src)
        assert m, pattern
        return m
    assert abs(p["time_step"] - 1.0 / 60) < 1e-15 and "self.timeStep=1./60" in src
    assert p["n_substeps"] == int(grab(r"self\.n_substeps = (\d+)").group(1))
    assert p["distance_threshold"] == float(grab(r"self\.distance_threshold=([\d.]+)").group(1))
    assert p["max_vel"] == float(grab(r"self\.max_vel = ([\d.]+)").group(1))
    assert p["max_gripper_vel"] == float(grab(r"self\.max_gripper_vel = ([\d.]+)").group(1))
    assert p["max_episode_steps"] == int(grab(r"self\._max_episode_steps = (\d+)").group(1))
    assert p["action_dt"] == p["time_step"] * p["n_substeps"]          # self.dt (:28)
    m = grab(r"self\.pos_space = spaces\.Box\(low=np\.array\(\[([^\]]+)\]\), high=np\.array\(\[([^\]]+)\]\)")
    assert [float(x) for x in m.group(1).split(",")] == p["pos_low"] and [float(x) for x in m.group(2).split(",")] == p["pos_high"]
    m = grab(r"self\.goal_space = spaces\.Box\(low=np\.array\(\[([^\]]+)\]\),high=np\.array\(\[([^\]]+)\]\)")
    assert [float(x) for x in m.group(1).split(",")] == p["goal_low"] and [float(x) for x in m.group(2).split(",")] == p["goal_high"]
    m = grab(r"self\.obj_space = spaces\.Box\(low=np\.array\(\[([^\]]+)\]\), high=np\.array\(\[([^\]]+)\]\)")
    assert [float(x) for x in m.group(1).split(",")] == p["obj_low"] and [float(x) for x in m.group(2).split(",")] == p["obj_high"]
    m = grab(r"self\.gripper_space = spaces\.Box\(low=([\d.]+), high=([\d.]+)")
    assert float(m.group(1)) == p["gripper_low"] and float(m.group(2)) == p["gripper_high"]
    m = grab(r"self\.startGripperPos = \[([^\]]+)\]")
    assert [float(x) for x in m.group(1).split(",")] == p["start_gripper_pos"]
    assert "halfExtents = [self.lego_length/2, 0.025, 0.04]" in src and p["obj_half"] == [0.025, 0.025, 0.04]
    assert "baseMass = 0.5" in src and p["obj_mass"] == 0.5
    assert "force=1000" in src and p["finger_motor_force"] == 1000.0
    assert "gearRatio=-1, erp=0.1, maxForce=50" in src
    assert JS["solver"]["gear_erp"] == 0.1 and JS["solver"]["gear_max_force"] == 50.0
    assert "lateralFriction = 100" in src and JS["solver"]["mu_finger_grasp"] == 100.0


def test_primitives_urdf_round_trips_the_model_table():
    """tools/emit_primitives_urdf.py: the URDF a PyBullet side-by-side run would load carries exactly the model table's
    joints, masses, inertias and the build's collision primitives"""
    import json
    import sys
    import xml.etree.ElementTree as ET
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from emit_primitives_urdf import emit
    js = json.load(open(os.path.join(ROOT, "gym_xarm_amd", "model", "xarm7_pd.json")))
    root = ET.fromstring(emit(js))
    links = {l.attrib["name"]: l for l in root.findall("link")}
    joints = {j.find("child").attrib["link"]: j for j in root.findall("joint")}
    assert len(links) == len(js["links"]) + 1 and len(joints) == len(js["links"])
    for i, l in enumerate(js["links"]):
        e, j = links[l["name"]], joints[l["name"]]
        assert float(e.find("inertial/mass").attrib["value"]) == l["mass"]
        assert [float(x) for x in e.find("inertial/origin").attrib["xyz"].split()] == [float(x) for x in l["com"]]
        assert float(e.find("inertial/inertia").attrib["ixx"]) == l["inertia"][0] and float(e.find("inertial/inertia").attrib["izz"]) == l["inertia"][5]
        assert j.attrib["type"] == l["joint"]
        assert [float(x) for x in j.find("origin").attrib["xyz"].split()] == [float(x) for x in l["origin_xyz"]]
        if l["joint"] != "fixed":
            assert [float(x) for x in j.find("axis").attrib["xyz"].split()] == [float(x) for x in l["axis"]]
            assert float(j.find("limit").attrib["lower"]) == l["lower"] and float(j.find("dynamics").attrib["damping"]) == l["damping"]
        n_col = len(e.findall("collision"))
        assert n_col == (len(js["pads"]["centers_left"]) if i in js["finger_links"] else 0)     # only the pad spheres collide


def test_pybullet_harness_reports_unavailable_cleanly():
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "pybullet_harness.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    assert out.stdout.startswith("reference: unavailable (pybullet not importable") or out.stdout.startswith("reference: pybullet")
