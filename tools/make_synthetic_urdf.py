#!/usr/bin/env python3
"""Writes wbc_quadruped_dob_amd/assets/synthetic_quadruped.urdf.

SYNTHETIC stand-in: the real DogBot URDF
(DogBotV4/ROS/src/dogbot_description/urdf/dogbot.urdf, /root/reference/README.md:60)
lives in an un-vendored submodule and is absent.  Every mass, length and inertia
below is invented (roughly DogBot-sized: 4 legs x {roll, pitch, knee}); nothing here
is a fact about the reference robot.  Deliberately awkward numbers (rotated joint
origins, rotated inertial frames, off-axis COMs, products of inertia, mirrored axes,
fixed-joint foot links that must be lumped) so that loaders and dynamics are
exercised in full generality.
"""
import os

LEGS = [  # name, x, y, side(+1 left / -1 right)
    ("front_left", 0.28, 0.12, 1.0),
    ("front_right", 0.28, -0.12, -1.0),
    ("back_left", -0.28, 0.12, 1.0),
    ("back_right", -0.28, -0.12, -1.0),
]


def inertial(m, xyz, rpy, ixx, ixy, ixz, iyy, iyz, izz):
    return (f'    <inertial>\n      <origin xyz="{xyz}" rpy="{rpy}"/>\n      <mass value="{m}"/>\n'
            f'      <inertia ixx="{ixx}" ixy="{ixy}" ixz="{ixz}" iyy="{iyy}" iyz="{iyz}" izz="{izz}"/>\n'
            f'    </inertial>\n')


def main():
    o = ['<?xml version="1.0"?>\n',
         '<!-- SYNTHETIC quadruped (NOT the DogBot URDF, which is absent): see tools/make_synthetic_urdf.py -->\n',
         '<robot name="synthetic_quadruped">\n',
         '  <link name="base_link">\n',
         inertial(11.5, "0.012 -0.003 0.018", "0.02 -0.01 0.03", 0.118, 0.0021, -0.0043, 0.352, 0.0011, 0.401),
         '  </link>\n',
         # a fixed-joint payload that has to be lumped into the base
         '  <link name="payload">\n',
         inertial(0.8, "0.0 0.0 0.01", "0 0 0", 0.004, 0.0, 0.0, 0.006, 0.0, 0.008),
         '  </link>\n',
         '  <joint name="payload_mount" type="fixed">\n    <parent link="base_link"/>\n    <child link="payload"/>\n'
         '    <origin xyz="-0.05 0.01 0.06" rpy="0 0.1 0"/>\n  </joint>\n']
    for name, x, y, s in LEGS:
        fr = 1.0 if x > 0 else -1.0
        # hip roll
        o += [f'  <link name="{name}_hip">\n',
              inertial(1.15, f"{0.004*fr} {0.021*s} -0.002", "0.0 0.0 0.0", 0.0021, 0.0001 * s, 0.0, 0.0026, 0.0, 0.0019),
              '  </link>\n',
              f'  <joint name="{name}_roll_joint" type="revolute">\n    <parent link="base_link"/>\n'
              f'    <child link="{name}_hip"/>\n    <origin xyz="{x} {y} -0.01" rpy="0 0 {0.04*s*fr}"/>\n'
              f'    <axis xyz="{1.0*fr} 0 0"/>\n    <limit lower="-1.0" upper="1.0" effort="60" velocity="20"/>\n  </joint>\n']
        # hip pitch / upper leg
        o += [f'  <link name="{name}_upper">\n',
              inertial(1.62, f"-0.006 {0.012*s} -0.118", f"0.0 {0.03*s} 0.0", 0.0171, 0.0, 0.0004 * s, 0.0178, 0.0002, 0.0017),
              '  </link>\n',
              f'  <joint name="{name}_pitch_joint" type="revolute">\n    <parent link="{name}_hip"/>\n'
              f'    <child link="{name}_upper"/>\n    <origin xyz="0.0 {0.062*s} 0.0" rpy="{0.02*s} 0 0"/>\n'
              f'    <axis xyz="0 1 0"/>\n    <limit lower="-2.5" upper="2.5" effort="60" velocity="20"/>\n  </joint>\n']
        # knee / lower leg
        o += [f'  <link name="{name}_lower">\n',
              inertial(0.44, "0.005 0.0 -0.135", "0 0 0", 0.0041, 0.0, 0.0001, 0.0042, 0.0, 0.0003),
              '  </link>\n',
              f'  <joint name="{name}_knee_joint" type="revolute">\n    <parent link="{name}_upper"/>\n'
              f'    <child link="{name}_lower"/>\n    <origin xyz="0.0 0.0 -0.30" rpy="0 0 0"/>\n'
              f'    <axis xyz="0 {1.0 if s > 0 else -1.0} 0"/>\n    <limit lower="-2.6" upper="2.6" effort="60" velocity="20"/>\n  </joint>\n']
        # foot: fixed link lumped into the lower leg, its origin is the contact point
        o += [f'  <link name="{name}_foot">\n',
              inertial(0.06, "0 0 0.005", "0 0 0", 0.00002, 0.0, 0.0, 0.00002, 0.0, 0.00002),
              '  </link>\n',
              f'  <joint name="{name}_foot_joint" type="fixed">\n    <parent link="{name}_lower"/>\n'
              f'    <child link="{name}_foot"/>\n    <origin xyz="0.01 0.0 -0.31" rpy="0 0 0"/>\n  </joint>\n']
    o.append('</robot>\n')
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wbc_quadruped_dob_amd", "assets",
                       "synthetic_quadruped.urdf")
    with open(out, "w") as f:
        f.write("".join(o))
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
