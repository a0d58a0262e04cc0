#!/usr/bin/env python3
"""Emit a primitives-only URDF of the model the kernels simulate (gym_xarm_amd/model/xarm7_pd.json) - SURVEY.md 7
steps 1 and 8: PyBullet, where it is available, can then simulate the IDENTICAL model (same joint frames, masses,
inertias, and the build's collision primitives: two pad spheres per finger, no arm / hand collision geometry) side by
side with the kernels (tests/tools/pybullet_harness.py).  Nothing here reads the reference's URDF or meshes; the numbers
come from the model table, whose entries cite them.

  python tools/emit_primitives_urdf.py [out.urdf]
"""
import json
import os
import sys
from xml.sax.saxutils import quoteattr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _v(xs):
    return " ".join(repr(float(x)) for x in xs)


def emit(js):
    links, pads = js["links"], js["pads"]
    out = ['<?xml version="1.0"?>', "<robot name=%s>" % quoteattr(js["name"] + "_primitives"), '  <link name="link_base"/>']
    finger = set(js["finger_links"])
    for i, l in enumerate(links):
        ixx, ixy, ixz, iyy, iyz, izz = l["inertia"]
        out.append("  <link name=%s>" % quoteattr(l["name"]))
        out.append('    <inertial><origin xyz="%s" rpy="0 0 0"/><mass value="%r"/>' % (_v(l["com"]), float(l["mass"])))
        out.append('      <inertia ixx="%r" ixy="%r" ixz="%r" iyy="%r" iyz="%r" izz="%r"/></inertial>' % (ixx, ixy, ixz, iyy, iyz, izz))
        if i in finger:
            sign = 1.0 if l["axis"][1] > 0 else -1.0   # right finger: pad centres mirrored in y
            for c in pads["centers_left"]:
                out.append('    <collision><origin xyz="%s" rpy="0 0 0"/><geometry><sphere radius="%r"/></geometry></collision>'
                           % (_v([c[0], sign * c[1], c[2]]), float(pads["radius"])))
            out.append('    <contact><lateral_friction value="1.0"/><stiffness value="%r"/><damping value="%r"/></contact>'
                       % (float(js["solver"]["finger_contact_stiffness"]), float(js["solver"]["finger_contact_damping"])))
        out.append("  </link>")
        parent = "link_base" if l["parent"] < 0 else links[l["parent"]]["name"]
        out.append("  <joint name=%s type=%s>" % (quoteattr("joint_" + l["name"]), quoteattr(l["joint"])))
        out.append('    <parent link=%s/><child link=%s/>' % (quoteattr(parent), quoteattr(l["name"])))
        out.append('    <origin xyz="%s" rpy="%s"/>' % (_v(l["origin_xyz"]), _v(l["origin_rpy"])))
        if l["joint"] != "fixed":
            out.append('    <axis xyz="%s"/><limit lower="%r" upper="%r" effort="100" velocity="10"/><dynamics damping="%r" friction="0"/>'
                       % (_v(l["axis"]), float(l["lower"]), float(l["upper"]), float(l["damping"])))
        out.append("  </joint>")
    out.append("</robot>")
    return "\n".join(out) + "\n"


def main():
    js = json.load(open(os.path.join(ROOT, "gym_xarm_amd", "model", "xarm7_pd.json")))
    text = emit(js)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
