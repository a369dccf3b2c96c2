"""TEST INFRASTRUCTURE (oracle side) -- URDF subset -> flat model arrays, in Python.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product has its own C++ reader (wbc_quadruped_dob_amd/csrc/urdf_reader.cpp); the two are
written independently (this one on xml.etree, that one on a hand-written tokenizer) and
tests/test_urdf_reader.py checks that they produce the same flat model.

Reference anchor: the controller takes the URDF path as argv[1]
(/root/reference/README.md:60).  The reference's own model loader is in an absent
submodule (/root/reference/.gitmodules:4-6), so the flat layout below is this build's.

Flat model (all float64, row-major):
  nb            bodies; body 0 = floating base, body i>=1 = child link of movable joint i
                (depth-first order from the root, children in document order); q index = i-1
  parent[nb]    parent body (-1 for the base); parent[i] < i
  Rt[nb,9]      rotation child-joint-frame(q=0) -> parent frame   (identity for body 0)
  rt[nb,3]      joint origin in parent coordinates
  axis[nb,3]    unit joint axis in the child frame (zeros for body 0)
  mass[nb], com[nb,3], Ic[nb,6]   link inertia incl. lumped fixed-joint children;
                Ic = (xx,xy,xz,yy,yz,zz) about the COM, in link axes
  foot_body[nf], foot_off[nf,3]   contact points: body and offset in that body's frame
  gravity[3]
"""
import xml.etree.ElementTree as ET
import numpy as np


def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    # URDF fixed-axis roll-pitch-yaw: R = Rz(y) Ry(p) Rx(r)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _vec(s, n=3):
    v = [float(t) for t in s.split()]
    assert len(v) == n
    return np.array(v)


class _Link:
    def __init__(self):
        self.m = 0.0
        self.c = np.zeros(3)
        self.I = np.zeros((3, 3))  # about COM, link axes


def _lump(a, b_m, b_c, b_I):
    """lump inertia (b_m, b_c, b_I about its COM) into _Link a; everything in a's frame."""
    if b_m == 0.0:
        return
    m = a.m + b_m
    c = (a.m * a.c + b_m * b_c) / m
    I = np.zeros((3, 3))
    for mk, ck, Ik in ((a.m, a.c, a.I), (b_m, b_c, b_I)):
        d = ck - c
        I += Ik + mk * (d.dot(d) * np.eye(3) - np.outer(d, d))
    a.m, a.c, a.I = m, c, I


def load_urdf(path, foot_links=None, gravity=(0.0, 0.0, -9.81)):
    root = ET.parse(path).getroot()
    links = {}
    order = []
    for L in root.findall("link"):
        lk = _Link()
        ine = L.find("inertial")
        if ine is not None:
            org = ine.find("origin")
            xyz = _vec(org.get("xyz", "0 0 0")) if org is not None else np.zeros(3)
            rpy = _vec(org.get("rpy", "0 0 0")) if org is not None else np.zeros(3)
            Ri = rpy_to_R(*rpy)
            lk.m = float(ine.find("mass").get("value"))
            it = ine.find("inertia")
            g = lambda k: float(it.get(k, "0"))
            I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
            lk.c = xyz
            lk.I = Ri @ I @ Ri.T
        links[L.get("name")] = lk
        order.append(L.get("name"))
    joints = []
    for J in root.findall("joint"):
        org = J.find("origin")
        xyz = _vec(org.get("xyz", "0 0 0")) if org is not None else np.zeros(3)
        rpy = _vec(org.get("rpy", "0 0 0")) if org is not None else np.zeros(3)
        ax = J.find("axis")
        axis = _vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0.0, 0.0])
        t = J.get("type")
        if t not in ("revolute", "continuous", "fixed"):
            raise ValueError("unsupported joint type %s" % t)
        joints.append(dict(name=J.get("name"), type=t, parent=J.find("parent").get("link"),
                           child=J.find("child").get("link"), R=rpy_to_R(*rpy), r=xyz,
                           axis=axis / np.linalg.norm(axis)))
    children = {j["child"] for j in joints}
    roots = [n for n in order if n not in children]
    assert len(roots) == 1, roots
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j["parent"], []).append(j)

    bodies = []  # dict(link, parent, Rt, rt, axis, lk(_Link), name)
    frames = {}  # link name -> (body index, R link->body, r link origin in body)
    joint_names = []

    def visit(link, body, R_lb, r_lb):
        frames[link] = (body, R_lb, r_lb)
        lk = links[link]
        _lump(bodies[body]["lk"], lk.m, r_lb + R_lb @ lk.c, R_lb @ lk.I @ R_lb.T)
        for j in by_parent.get(link, []):
            if j["type"] == "fixed":
                visit(j["child"], body, R_lb @ j["R"], r_lb + R_lb @ j["r"])
            else:
                bodies.append(dict(parent=body, Rt=R_lb @ j["R"], rt=r_lb + R_lb @ j["r"], axis=j["axis"],
                                   lk=_Link(), name=j["child"]))
                joint_names.append(j["name"])
                visit(j["child"], len(bodies) - 1, np.eye(3), np.zeros(3))

    bodies.append(dict(parent=-1, Rt=np.eye(3), rt=np.zeros(3), axis=np.zeros(3), lk=_Link(), name=roots[0]))
    visit(roots[0], 0, np.eye(3), np.zeros(3))

    nb = len(bodies)
    if foot_links is None:
        # default: the last link (document order) hanging off each leaf body
        leaf = [i for i in range(1, nb) if all(b["parent"] != i for b in bodies)]
        foot_links = []
        for i in leaf:
            cands = [n for n in order if frames[n][0] == i]
            foot_links.append(cands[-1])
    fb, fo = [], []
    for n in foot_links:
        b, _, r = frames[n]
        fb.append(b)
        fo.append(r)
    M = dict(
        nb=nb,
        parent=np.array([b["parent"] for b in bodies], dtype=np.int32),
        Rt=np.array([b["Rt"].reshape(9) for b in bodies]),
        rt=np.array([b["rt"] for b in bodies]),
        axis=np.array([b["axis"] for b in bodies]),
        mass=np.array([b["lk"].m for b in bodies]),
        com=np.array([b["lk"].c for b in bodies]),
        Ic=np.array([[b["lk"].I[0, 0], b["lk"].I[0, 1], b["lk"].I[0, 2], b["lk"].I[1, 1], b["lk"].I[1, 2],
                      b["lk"].I[2, 2]] for b in bodies]),
        foot_body=np.array(fb, dtype=np.int32),
        foot_off=np.array(fo).reshape(-1, 3),
        gravity=np.array(gravity, dtype=np.float64),
        joint_names=joint_names,
        foot_links=list(foot_links),
        body_names=[b["name"] for b in bodies],
    )
    return M
