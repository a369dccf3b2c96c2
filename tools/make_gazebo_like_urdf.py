#!/usr/bin/env python3
"""Writes tests/golden/gazebo_like_quadruped.urdf: a SYNTHETIC quadruped (NOT the DogBot URDF, which is absent) dressed
the way a Gazebo / ros_control robot description is, to stress the URDF readers with what a real `dogbot.urdf` is likely
to contain (/root/reference/README.md:14,38,60: Gazebo model, ros-control): xacro namespace and comments, <material>,
<gazebo> blocks with plugins, <transmission>, mesh <visual>/<collision> with their own origins, <dynamics>,
<safety_controller>, <mimic>, `continuous` knees, joint axes that are neither axis-aligned nor normalised, links declared
after the joints that use them, legs interleaved and back to front, fixed sensor / guard links on the trunk and on leg
bodies (to be lumped), scientific notation, tabs and newlines inside vectors, single-quoted attributes.
Numbers are invented.  Feet must be named explicitly (sensor links follow the feet in document order)."""
import os

LEGS = [("fl", 0.31, 0.11, 1.0), ("fr", 0.31, -0.11, -1.0), ("rl", -0.27, 0.13, 1.0), ("rr", -0.27, -0.13, -1.0)]
FEET = ["fl_foot", "fr_foot", "rl_foot", "rr_foot"]


def inertial(m, xyz, rpy, I):
    return ('    <inertial>\n      <mass value="%s"/>\n      <origin rpy="%s" xyz="%s"/>\n'
            '      <inertia ixx="%s" ixy="%s" ixz="%s" iyy="%s" iyz="%s" izz="%s"/>\n    </inertial>\n' % ((m, rpy, xyz) + tuple(I)))


def visual(mesh, xyz="0 0 0"):
    return ('    <visual>\n      <origin xyz="%s" rpy="0 0 1.5708"/>\n      <geometry><mesh filename="package://dogbot_description/meshes/%s.dae" scale="0.001 0.001 0.001"/></geometry>\n'
            '      <material name="grey"/>\n    </visual>\n'
            '    <collision>\n      <origin xyz="%s" rpy="0 0 0"/>\n      <geometry><box size="0.1 0.05 0.03"/></geometry>\n    </collision>\n' % (xyz, mesh, xyz))


def leg(name, x, y, s):
    fr = 1.0 if x > 0 else -1.0
    links = [
        '  <link name="%s_hip">\n%s%s  </link>\n' % (name, visual("hip"), inertial("1.21", "%g %g -1e-3" % (0.005 * fr, 0.019 * s), "0.01 0 0", ("2.3e-3", 1e-4 * s, 0, "2.9E-3", 0, 0.0021))),
        '  <link name="%s_upper">\n%s%s  </link>\n' % (name, inertial(1.55, "-0.004 %g -0.121" % (0.011 * s), "0 %g 0" % (0.02 * s), (0.0169, 0, 0.0003 * s, 0.0175, 0.0002, 0.0019)), visual("upper", "0 0 -0.15")),
        # a guard with mass, fixed to the upper leg: lumped into it
        "  <link name='%s_shin_guard'>\n%s  </link>\n" % (name, inertial(0.09, "0.01 0 -0.02", "0 0 0", (4e-5, 0, 0, 5e-5, 0, 3e-5))),
        '  <link name="%s_lower">\n%s%s  </link>\n' % (name, visual("lower", "0 0 -0.14"), inertial(0.41, "0.004 0 -0.131", "0 0 0", (0.0039, 0, 0.0001, 0.004, 0, 0.0003))),
        '  <link name="%s_foot">\n%s    <collision><geometry><sphere radius="0.02"/></geometry></collision>\n  </link>\n' % (name, inertial(0.055, "0 0 0.004", "0 0 0", (2e-5, 0, 0, 2e-5, 0, 2e-5))),
        # massless contact-sensor frame on the lower leg, declared AFTER the foot
        '  <link name="%s_contact_sensor"/>\n' % name,
    ]
    joints = [
        '  <joint name="%s_roll" type="revolute">\n    <origin xyz="%g %g -0.012" rpy="0 0 %g"/>\n    <parent link="trunk"/>\n    <child link="%s_hip"/>\n'
        '    <axis xyz="%g 0.02 -0.015"/>\n    <limit lower="-0.9" upper="0.9" effort="55" velocity="18"/>\n    <dynamics damping="0.1" friction="0.05"/>\n'
        '    <safety_controller soft_lower_limit="-0.85" soft_upper_limit="0.85" k_position="20" k_velocity="5"/>\n  </joint>\n' % (name, x, y, 0.03 * s * fr, name, 2.0 * fr),
        '  <joint name="%s_pitch" type="revolute">\n    <parent link="%s_hip"/>\n    <child link="%s_upper"/>\n    <origin xyz="0 %g 0"\n            rpy="%g\t0   0"/>\n'
        '    <axis xyz="0.05 0.99 -0.03"/>\n    <limit effort="55" velocity="18" lower="-2.4" upper="2.4"/>\n  </joint>\n' % (name, name, name, 0.06 * s, 0.015 * s),
        '  <joint name="%s_guard_mount" type="fixed">\n    <parent link="%s_upper"/>\n    <child link="%s_shin_guard"/>\n    <origin xyz="0.02 0 -0.2" rpy="0 0.05 0"/>\n  </joint>\n' % (name, name, name),
        # continuous knee (no limits), axis mirrored left/right and not normalised; the <mimic> element is ignored by the readers
        '  <joint name="%s_knee" type="continuous">\n    <parent link="%s_upper"/>\n    <child link="%s_lower"/>\n    <origin xyz="0 0 -0.29"/>\n'
        '    <axis xyz="0 %g 0.06"/>\n    <mimic joint="%s_pitch" multiplier="-1.0" offset="0.0"/>\n  </joint>\n' % (name, name, name, 3.0 * s, name),
        '  <joint name="%s_ankle" type="fixed">\n    <parent link="%s_lower"/>\n    <child link="%s_foot"/>\n    <origin xyz="0.012 0 -0.305" rpy="0 0 0"/>\n  </joint>\n' % (name, name, name),
        '  <joint name="%s_contact_mount" type="fixed">\n    <parent link="%s_lower"/>\n    <child link="%s_contact_sensor"/>\n    <origin xyz="0.012 0 -0.32"/>\n  </joint>\n' % (name, name, name),
    ]
    extras = [
        '  <transmission name="%s_%s_tran">\n    <type>transmission_interface/SimpleTransmission</type>\n    <joint name="%s_%s"><hardwareInterface>hardware_interface/EffortJointInterface</hardwareInterface></joint>\n'
        '    <actuator name="%s_%s_motor"><mechanicalReduction>1</mechanicalReduction></actuator>\n  </transmission>\n' % (name, j, name, j, name, j) for j in ("roll", "pitch", "knee")
    ] + ['  <gazebo reference="%s_foot">\n    <mu1>0.6</mu1><mu2>0.6</mu2><kp value="1e6"/><kd value="1.0"/>\n    <sensor name="%s_contact" type="contact"><contact><collision>%s_foot_collision</collision></contact></sensor>\n  </gazebo>\n' % (name, name, name)]
    return links, joints, extras


def main():
    head = ['<?xml version="1.0" encoding="utf-8"?>\n',
            '<!-- SYNTHETIC Gazebo-style quadruped for reader stress tests (NOT the DogBot URDF): tools/make_gazebo_like_urdf.py -->\n',
            '<robot name="gazebo_like_quadruped" xmlns:xacro="http://www.ros.org/wiki/xacro">\n',
            '  <material name="grey"><color rgba="0.5 0.5 0.5 1"/></material>\n',
            '  <gazebo>\n    <plugin name="gazebo_ros_control" filename="libgazebo_ros_control.so">\n      <robotNamespace>/dogbot</robotNamespace>\n    </plugin>\n  </gazebo>\n']
    trunk = ['  <link name="trunk">\n' + visual("body") + inertial(12.1, "0.01 -0.002 0.02", "0.01 -0.02 0.02", (0.121, 0.002, -0.004, 0.36, 0.001, 0.41)) + '  </link>\n',
             '  <link name="imu_link">\n' + inertial(0.02, "0 0 0", "0 0 0", (1e-6, 0, 0, 1e-6, 0, 1e-6)) + '  </link>\n',
             '  <joint name="imu_joint" type="fixed">\n    <parent link="trunk"/>\n    <child link="imu_link"/>\n    <origin xyz="0.05 0 0.04" rpy="0 0 3.14159"/>\n  </joint>\n',
             '  <gazebo reference="imu_link">\n    <sensor name="imu" type="imu"><always_on>true</always_on><update_rate>500</update_rate></sensor>\n  </gazebo>\n']
    parts = {n: leg(n, x, y, s) for n, x, y, s in LEGS}
    body = []
    # joints of all legs first (links they name come later), legs back to front and interleaved
    for n in ("rr", "fl", "rl", "fr"):
        body += parts[n][1]
    body += trunk
    for n in ("fr", "rl", "rr", "fl"):
        body += parts[n][0] + parts[n][2]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "gazebo_like_quadruped.urdf")
    with open(out, "w") as f:
        f.write("".join(head + body) + "</robot>\n")
    print("wrote", os.path.normpath(out), "feet:", FEET)


if __name__ == "__main__":
    main()
