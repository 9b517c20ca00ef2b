"""Model compiler: robot.mjcf + metadata.json -> kbj_model blob.

Replaces, for this one robot family, what the reference obtains from MuJoCo's XML compiler
(train.py:1079-1081 mujoco_scenes.mjcf.load_mjmodel -> mujoco.MjModel) and from ksim's metadata
loader (train.py:1083-1089). The floor is an infinite plane z = 0 added programmatically
(the reference gets geom "floor" from mujoco_scenes, train.py:1081,1113).

Only the MJCF features this robot uses are understood: nested <default class>, childclass,
<inertial> with diagonal inertia, hinge/free joints, capsule geoms given by fromto, box sites.
Anything else raises, rather than being silently ignored.
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import constants as K
from .layout import MAGIC, NBODY, NCAP, NQ, NU, NV, VERSION, Model

DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
GEOM_DENSITY = 1000.0  # MuJoCo default density for bodies without <inertial>


def _floats(s: str) -> list[float]:
    return [float(x) for x in s.split()]


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


class _Defaults:
    """<default class=...> tree flattened to {class: {tag: attrs}} with inheritance."""

    def __init__(self, root: ET.Element):
        self.cls: dict[str, dict[str, dict[str, str]]] = {"main": {}}
        top = root.find("default")
        if top is not None:
            self._walk(top, "main", {})

    def _walk(self, node: ET.Element, name: str, inherited: dict):
        cur = {k: dict(v) for k, v in inherited.items()}
        for ch in node:
            if ch.tag != "default":
                cur.setdefault(ch.tag, {}).update(ch.attrib)
        self.cls[name] = cur
        for ch in node:
            if ch.tag == "default":
                self._walk(ch, ch.attrib["class"], cur)

    def get(self, cls: str | None, tag: str) -> dict[str, str]:
        return dict(self.cls.get(cls or "main", self.cls["main"]).get(tag, {}))


def compile_model(robot_dir: str) -> Model:
    """Compile `<robot_dir>/robot.mjcf` + `<robot_dir>/metadata.json` into a kbj_model."""
    tree = ET.parse(os.path.join(robot_dir, "robot.mjcf"))
    root = tree.getroot()
    comp = root.find("compiler")
    if comp is not None and comp.attrib.get("angle", "degree") != "radian":
        raise ValueError("only angle=radian MJCFs are supported")
    dfl = _Defaults(root)
    with open(os.path.join(robot_dir, "metadata.json")) as f:
        meta = json.load(f)

    bodies: list[dict] = [dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), joint=None,
                               inertial=None, geoms=[], sites=[])]

    def walk(elem: ET.Element, parent: int, childclass: str | None):
        cc = elem.attrib.get("childclass", childclass)
        b = dict(name=elem.attrib["name"], parent=parent,
                 pos=np.array(_floats(elem.attrib.get("pos", "0 0 0"))),
                 quat=np.array(_floats(elem.attrib.get("quat", "1 0 0 0"))), joint=None, inertial=None, geoms=[],
                 sites=[])
        b["quat"] = b["quat"] / np.linalg.norm(b["quat"])
        idx = len(bodies)
        bodies.append(b)
        for ch in elem:
            if ch.tag == "freejoint":
                b["joint"] = dict(type="free", name=ch.attrib.get("name", ""))
            elif ch.tag == "joint":
                a = dfl.get(ch.attrib.get("class", cc), "joint")
                a.update(ch.attrib)
                if a.get("type", "hinge") != "hinge":
                    raise ValueError(f"unsupported joint type {a.get('type')}")
                if any(abs(x) > 0 for x in _floats(a.get("pos", "0 0 0"))) or float(a.get("ref", 0.0)) != 0.0:
                    raise ValueError("joint pos/ref offsets are not supported")
                if b["joint"] is not None:
                    raise ValueError("more than one joint per body is not supported")
                b["joint"] = dict(type="hinge", name=a["name"], axis=np.array(_floats(a.get("axis", "0 0 1"))),
                                  range=_floats(a["range"]), armature=float(a.get("armature", 0)),
                                  frictionloss=float(a.get("frictionloss", 0)), damping=float(a.get("damping", 0)),
                                  frcrange=_floats(a.get("actuatorfrcrange", "0 0")))
                if b["joint"]["damping"] != 0.0:
                    raise ValueError("joint damping is not supported (the robot defines none)")
            elif ch.tag == "inertial":
                q = _floats(ch.attrib.get("quat", "1 0 0 0"))
                if not np.allclose(np.abs(q), [1, 0, 0, 0]):
                    raise ValueError("rotated inertial frames are not supported")
                b["inertial"] = dict(pos=np.array(_floats(ch.attrib["pos"])), mass=float(ch.attrib["mass"]),
                                     diag=np.array(_floats(ch.attrib["diaginertia"])))
            elif ch.tag == "geom":
                a = dfl.get(ch.attrib.get("class", cc), "geom")
                a.update(ch.attrib)
                b["geoms"].append(a)
            elif ch.tag == "site":
                b["sites"].append(dict(ch.attrib))
            elif ch.tag == "body":
                walk(ch, idx, cc)
            elif ch.tag == "camera":
                pass
            else:
                raise ValueError(f"unsupported element <{ch.tag}> in body {b['name']}")

    for top in root.find("worldbody"):
        if top.tag == "body":
            walk(top, 0, None)
    if len(bodies) != NBODY:
        raise ValueError(f"expected {NBODY} bodies (kbot topology), found {len(bodies)}")

    m = Model()
    m.magic, m.version = MAGIC, VERSION
    m.nbody, m.nq, m.nv, m.nu, m.ncap = NBODY, NQ, NV, NU, NCAP
    name2body = {b["name"]: i for i, b in enumerate(bodies)}

    # ---- tree, joints, dofs ----
    dof = 0
    qadr = 0
    last_dof_of_body = [-1] * NBODY
    joint_names: list[str] = []
    qpos0 = []
    for i, b in enumerate(bodies):
        m.body_parent[i] = b["parent"]
        for k in range(3):
            m.body_pos[i][k] = b["pos"][k]
        for k in range(4):
            m.body_quat[i][k] = b["quat"][k]
        j = b["joint"]
        # nearest ancestor dof
        anc = b["parent"]
        pdof = -1
        while anc > 0 and last_dof_of_body[anc] < 0:
            anc = bodies[anc]["parent"]
        if anc > 0:
            pdof = last_dof_of_body[anc]
        if j is None:
            m.body_dofadr[i], m.body_dofnum[i] = -1, 0
        elif j["type"] == "free":
            if i != 1 or b["parent"] != 0:
                raise ValueError("the free joint must be on body 1")
            m.body_dofadr[i], m.body_dofnum[i] = dof, 6
            for k in range(6):
                m.dof_body[dof + k] = i
                m.dof_parent[dof + k] = dof + k - 1 if k else pdof
            dof += 6
            last_dof_of_body[i] = dof - 1
            qpos0 += list(b["pos"]) + list(b["quat"])
            qadr += 7
        else:
            m.body_dofadr[i], m.body_dofnum[i] = dof, 1
            m.dof_body[dof], m.dof_parent[dof] = i, pdof
            ax = j["axis"] / np.linalg.norm(j["axis"])
            for k in range(3):
                m.jnt_axis[i][k] = ax[k]
            m.dof_armature[dof], m.dof_frictionloss[dof] = j["armature"], j["frictionloss"]
            m.dof_range[dof][0], m.dof_range[dof][1] = j["range"]
            joint_names.append(j["name"])
            last_dof_of_body[i] = dof
            dof += 1
            qpos0.append(0.0)
            qadr += 1
    if dof != NV or qadr != NQ:
        raise ValueError(f"expected nv={NV}, nq={NQ}; got {dof}, {qadr}")
    if tuple(joint_names) != K.JOINT_NAMES:
        raise ValueError("MJCF joint order does not match the task's JOINT_NAMES (train.py:22-23,70)")
    for k in range(NQ):
        m.qpos0[k] = qpos0[k]

    # ---- inertias ----
    for i, b in enumerate(bodies):
        if i == 0:
            continue
        if b["inertial"] is not None:
            ine = b["inertial"]
            mass, ipos, diag = ine["mass"], ine["pos"], ine["diag"]
        else:
            # MuJoCo infers mass from geoms when <inertial> is absent; the only such body (base) carries one
            # visual sphere (robot.mjcf:66 / kbot-headless robot.mjcf:68).
            spheres = [g for g in b["geoms"] if g.get("type") == "sphere"]
            if len(spheres) != 1 or len(b["geoms"]) != 1:
                raise ValueError(f"body {b['name']} has no <inertial> and is not a single sphere")
            r = float(spheres[0]["size"].split()[0])
            mass = GEOM_DENSITY * 4.0 / 3.0 * math.pi * r ** 3
            diag = np.full(3, 0.4 * mass * r * r)
            ipos = np.array(_floats(spheres[0].get("pos", "0 0 0")))
        m.body_mass[i] = mass
        for k in range(3):
            m.body_ipos[i][k] = ipos[k]
            m.body_inertia[i][k] = diag[k]

    # ---- actuators (motor per joint, same order) ----
    acts = root.find("actuator")
    motors = list(acts)
    if [mo.attrib["joint"] for mo in motors] != list(K.JOINT_NAMES):
        raise ValueError("actuator order must equal joint order")
    for u, mo in enumerate(motors):
        a = dfl.get(mo.attrib.get("class"), "motor")
        a.update(mo.attrib)
        lo, hi = _floats(a["ctrlrange"])
        m.act_range[u][0], m.act_range[u][1] = lo, hi
        jm = meta["joint_name_to_metadata"][K.JOINT_NAMES[u]]
        m.kp[u], m.kd[u], m.tau_limit[u] = float(jm["kp"]), float(jm["kd"]), float(jm["soft_torque_limit"])
        m.joint_bias[u] = K.JOINT_BIASES[u]
        m.joint_lo[u], m.joint_hi[u] = K.JOINT_LIMITS[u]

    # ---- collision capsules, sites ----
    caps = {}
    col = None
    for i, b in enumerate(bodies):
        for g in b["geoms"]:
            if g.get("name") in K.COLLISION_CAPSULES:
                caps[g["name"]] = (i, g)
            elif int(g.get("contype", 1)) != 0 or int(g.get("conaffinity", 1)) != 0:
                raise ValueError(f"unexpected colliding geom {g.get('name')}")
    for c, name in enumerate(K.COLLISION_CAPSULES):
        bi, g = caps[name]
        ft = np.array(_floats(g["fromto"]))
        p0, p1 = ft[:3], ft[3:]
        m.cap_body[c] = bi
        ax = (p1 - p0) / np.linalg.norm(p1 - p0)
        for k in range(3):
            m.cap_pos[c][k] = 0.5 * (p0[k] + p1[k])
            m.cap_axis[c][k] = ax[k]
        m.cap_halflen[c] = 0.5 * np.linalg.norm(p1 - p0)
        m.cap_radius[c] = float(g["size"].split()[0])
        col = g
    if int(col.get("condim", 3)) != 3:
        raise ValueError("only condim=3 contacts are supported")
    m.contact_mu = float(col["friction"].split()[0])  # capsule priority=1 > floor priority 0: capsule params win
    sr = _floats(col.get("solref", "0.02 1"))
    si = _floats(col.get("solimp", "0.9 0.95 0.001")) + [0.5, 2.0]
    m.contact_solref[0], m.contact_solref[1] = sr
    for k in range(5):
        m.contact_solimp[k] = si[k] if k < len(si) else DEFAULT_SOLIMP[k]
        m.limit_solimp[k] = DEFAULT_SOLIMP[k]
        m.fric_solimp[k] = DEFAULT_SOLIMP[k]
    m.limit_solref[0], m.limit_solref[1] = DEFAULT_SOLREF
    m.fric_solref[0], m.fric_solref[1] = DEFAULT_SOLREF

    m.base_body, m.lfoot_body, m.rfoot_body = name2body[K.BASE_BODY], name2body[K.FOOT_LEFT_BODY], name2body[K.FOOT_RIGHT_BODY]
    m.torso_body = 2
    site_owner = {}
    for i, b in enumerate(bodies):
        for s in b["sites"]:
            site_owner[s["name"]] = (i, s)
    for k, sname in enumerate(K.FOOT_SITES):
        bi, s = site_owner[sname]
        if bi != (m.lfoot_body, m.rfoot_body)[k] or s.get("type") != "box":
            raise ValueError("foot site must be a box on the foot body")
        for a in range(3):
            m.site_pos[k][a] = _floats(s["pos"])[a]
            m.site_size[k][a] = _floats(s["size"])[a]
    bi, s = site_owner[K.IMU_SITE]
    m.imu_body = bi
    iq = np.array(_floats(s.get("quat", "1 0 0 0")))
    if np.linalg.norm(_floats(s.get("pos", "0 0 0"))) != 0:
        raise ValueError("imu site offset not supported")
    for k in range(4):
        m.imu_quat[k] = iq[k] / np.linalg.norm(iq)
    m.gravity[0], m.gravity[1], m.gravity[2] = 0.0, 0.0, -9.81

    _set_const(m)
    return m


# ---------------------------------------------------------------------------------------------------
# constants that MuJoCo derives at compile time at qpos0: dof_invweight0, body_invweight0, meaninertia
# ---------------------------------------------------------------------------------------------------
def forward_kinematics(m: Model, qpos: np.ndarray):
    xpos = np.zeros((NBODY, 3))
    xquat = np.zeros((NBODY, 4))
    xquat[0, 0] = 1.0
    xmat = np.zeros((NBODY, 3, 3))
    xmat[0] = np.eye(3)
    qadr = 0
    for b in range(1, NBODY):
        p = m.body_parent[b]
        if m.body_dofnum[b] == 6:
            xpos[b] = qpos[0:3]
            xquat[b] = qpos[3:7] / np.linalg.norm(qpos[3:7])
            qadr = 7
        else:
            xpos[b] = xpos[p] + xmat[p] @ np.array(m.body_pos[b])
            q = quat_mul(xquat[p], np.array(m.body_quat[b]))
            if m.body_dofnum[b] == 1:
                ax = np.array(m.jnt_axis[b])
                ang = qpos[qadr]
                qadr += 1
                q = quat_mul(q, np.concatenate([[math.cos(ang / 2)], math.sin(ang / 2) * ax]))
            xquat[b] = q / np.linalg.norm(q)
        xmat[b] = quat_to_mat(xquat[b])
    return xpos, xquat, xmat


def body_jacobian(m: Model, xpos, xmat, body: int, point: np.ndarray):
    """(jacp, jacr) 3 x nv of a world point attached to `body` (free-joint rotation dofs are body-local)."""
    jp = np.zeros((3, NV))
    jr = np.zeros((3, NV))
    b = body
    while b > 0:
        n, adr = m.body_dofnum[b], m.body_dofadr[b]
        if n == 1:
            ax = xmat[b] @ np.array(m.jnt_axis[b])
            jr[:, adr] = ax
            jp[:, adr] = np.cross(ax, point - xpos[b])
        elif n == 6:
            jp[:, adr:adr + 3] = np.eye(3)
            for k in range(3):
                ax = xmat[b][:, k]
                jr[:, adr + 3 + k] = ax
                jp[:, adr + 3 + k] = np.cross(ax, point - xpos[b])
        b = m.body_parent[b]
    return jp, jr


def mass_matrix(m: Model, qpos: np.ndarray) -> np.ndarray:
    """Joint-space inertia by summing body Jacobian contributions (independent of the CRB formulation
    used by the oracle and the HIP kernel, so it cross-checks both)."""
    xpos, _, xmat = forward_kinematics(m, qpos)
    M = np.zeros((NV, NV))
    for b in range(1, NBODY):
        com = xpos[b] + xmat[b] @ np.array(m.body_ipos[b])
        jp, jr = body_jacobian(m, xpos, xmat, b, com)
        Iw = xmat[b] @ np.diag(np.array(m.body_inertia[b])) @ xmat[b].T
        M += m.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
    M += np.diag(np.array(m.dof_armature))
    return M


def _set_const(m: Model) -> None:
    qpos0 = np.array(m.qpos0, dtype=np.float64)
    M = mass_matrix(m, qpos0)
    Minv = np.linalg.inv(M)
    xpos, _, xmat = forward_kinematics(m, qpos0)
    m.meaninertia = float(np.trace(M) / NV)
    m.total_mass = float(sum(m.body_mass[b] for b in range(NBODY)))
    d = np.diag(Minv)
    for k in range(3):
        m.dof_invweight0[k] = float(d[0:3].mean())
        m.dof_invweight0[3 + k] = float(d[3:6].mean())
    for k in range(6, NV):
        m.dof_invweight0[k] = float(d[k])
    for b in range(1, NBODY):
        com = xpos[b] + xmat[b] @ np.array(m.body_ipos[b])
        jp, jr = body_jacobian(m, xpos, xmat, b, com)
        m.body_invweight0[b][0] = float(np.trace(jp @ Minv @ jp.T) / 3.0)
        m.body_invweight0[b][1] = float(np.trace(jr @ Minv @ jr.T) / 3.0)


# ---------------------------------------------------------------------------------------------------
# blob I/O
# ---------------------------------------------------------------------------------------------------
def model_to_bytes(m: Model) -> bytes:
    import ctypes
    return ctypes.string_at(ctypes.addressof(m), ctypes.sizeof(m))


def model_from_bytes(buf: bytes) -> Model:
    import ctypes
    if len(buf) != ctypes.sizeof(Model):
        raise ValueError(f"model blob has {len(buf)} bytes, expected {ctypes.sizeof(Model)}")
    m = Model.from_buffer_copy(buf)
    if m.magic != MAGIC or m.version != VERSION:
        raise ValueError("bad model blob magic/version")
    return m


_BLOB_DIR = os.path.join(os.path.dirname(__file__), "blobs")


def load_model(name: str = "kbot-headless") -> Model:
    """Load a committed, pre-compiled blob (the reference tree is not available on the GPU box)."""
    with open(os.path.join(_BLOB_DIR, f"{name}.kbjm"), "rb") as f:
        return model_from_bytes(f.read())


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description="compile robot.mjcf + metadata.json into a kbj model blob")
    ap.add_argument("robot_dir")
    ap.add_argument("out")
    a = ap.parse_args(argv)
    with open(a.out, "wb") as f:
        f.write(model_to_bytes(compile_model(a.robot_dir)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
