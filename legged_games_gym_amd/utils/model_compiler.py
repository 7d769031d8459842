"""URDF -> flat rigid-body tables ("model compiler").

Replaces what the reference obtains from Isaac Gym's asset importer
(``gym.load_asset`` + ``get_asset_*`` calls, reference
``legged_gym/envs/base/legged_robot.py:670-702``) with the asset options the
reference passes (``legged_robot_config.py:98-119``):

* ``collapse_fixed_joints=True``: links joined by a fixed joint are merged into
  the parent (mass, COM, inertia via the parallel-axis theorem, collision
  shapes re-expressed in the parent frame) unless the joint carries
  ``dont_collapse="true"`` (ANYmal-C feet, ``anymal_c.urdf:701``);
* ``replace_cylinder_with_capsule=True``: a URDF cylinder (axis = local z)
  becomes a capsule of the same radius whose segment spans the cylinder's
  length [EXTERNAL importer detail, assumption stated in DESIGN.md];
* body / DOF order: depth-first with alphabetically sorted siblings [EXTERNAL,
  matches the listing order in ``anymal_c_rough_config.py:44-57``].

The hot path needs the robot as *floating base + K identical-topology serial
limbs of L revolute joints* (ANYmal-C: K=4, L=3; Cassie: K=2, L=6); a body kept
only because of ``dont_collapse`` (a foot) stays a separate *report body* for
``contact_forces`` but is dynamically welded to its parent.  ``RobotModel``
holds both views; ``RobotModel.to_json`` / ``from_json`` give the compiled
table shipped under ``resources/models`` (so no URDF is needed at run time).
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


# ----------------------------------------------------------------------------- math
def rpy_to_matrix(rpy) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]], dtype=np.float64)


def axis_angle_matrix(axis, angle) -> np.ndarray:
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def _vec(text: Optional[str], n=3, default=0.0) -> np.ndarray:
    if text is None:
        return np.full(n, default, dtype=np.float64)
    return np.array([float(t) for t in text.split()], dtype=np.float64)


# ----------------------------------------------------------------------------- raw URDF
@dataclass
class _Shape:
    kind: str                 # sphere | capsule | box
    R: np.ndarray             # shape frame -> link frame
    p: np.ndarray
    radius: float = 0.0
    length: float = 0.0
    size: Optional[np.ndarray] = None
    owner: str = ""           # original link name


@dataclass
class _Link:
    name: str
    mass: float = 0.0
    com: np.ndarray = field(default_factory=lambda: np.zeros(3))
    inertia: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))   # at COM, link axes
    shapes: List[_Shape] = field(default_factory=list)


@dataclass
class _Joint:
    name: str
    kind: str
    parent: str
    child: str
    R: np.ndarray
    p: np.ndarray
    axis: np.ndarray
    lower: float = 0.0
    upper: float = 0.0
    has_limits: bool = False
    effort: float = 0.0
    velocity: float = 0.0
    damping: float = 0.0
    friction: float = 0.0
    dont_collapse: bool = False


def _parse_urdf(path: str):
    root = ET.parse(path).getroot()
    links: Dict[str, _Link] = {}
    for le in root.findall("link"):
        lk = _Link(le.get("name"))
        ie = le.find("inertial")
        if ie is not None:
            oe = ie.find("origin")
            Ri = rpy_to_matrix(_vec(oe.get("rpy") if oe is not None else None))
            lk.com = _vec(oe.get("xyz") if oe is not None else None)
            me = ie.find("mass")
            lk.mass = float(me.get("value")) if me is not None else 0.0
            te = ie.find("inertia")
            if te is not None:
                g = lambda k: float(te.get(k, "0"))
                I = np.array([[g("ixx"), g("ixy"), g("ixz")],
                              [g("ixy"), g("iyy"), g("iyz")],
                              [g("ixz"), g("iyz"), g("izz")]])
                lk.inertia = Ri @ I @ Ri.T
        for ce in le.findall("collision"):
            oe = ce.find("origin")
            R = rpy_to_matrix(_vec(oe.get("rpy") if oe is not None else None))
            p = _vec(oe.get("xyz") if oe is not None else None)
            ge = ce.find("geometry")
            if ge is None:
                continue
            if ge.find("sphere") is not None:
                lk.shapes.append(_Shape("sphere", R, p, radius=float(ge.find("sphere").get("radius")), owner=lk.name))
            elif ge.find("cylinder") is not None:
                c = ge.find("cylinder")
                lk.shapes.append(_Shape("cylinder", R, p, radius=float(c.get("radius")),
                                        length=float(c.get("length")), owner=lk.name))
            elif ge.find("capsule") is not None:
                c = ge.find("capsule")
                lk.shapes.append(_Shape("capsule", R, p, radius=float(c.get("radius")),
                                        length=float(c.get("length")), owner=lk.name))
            elif ge.find("box") is not None:
                lk.shapes.append(_Shape("box", R, p, size=_vec(ge.find("box").get("size")), owner=lk.name))
            # meshes: no in-scope robot collides with a mesh
        links[lk.name] = lk
    joints: List[_Joint] = []
    for je in root.findall("joint"):
        if je.find("parent") is None or je.find("child") is None:
            continue  # <transmission> children also use <joint name=.../>
        oe = je.find("origin")
        ae = je.find("axis")
        j = _Joint(je.get("name"), je.get("type"), je.find("parent").get("link"), je.find("child").get("link"),
                   rpy_to_matrix(_vec(oe.get("rpy") if oe is not None else None)),
                   _vec(oe.get("xyz") if oe is not None else None),
                   _vec(ae.get("xyz")) if ae is not None else np.array([1.0, 0.0, 0.0]))
        le = je.find("limit")
        if le is not None:
            j.effort = float(le.get("effort", "0"))
            j.velocity = float(le.get("velocity", "0"))
            if le.get("lower") is not None and le.get("upper") is not None:
                j.lower, j.upper, j.has_limits = float(le.get("lower")), float(le.get("upper")), True
        de = je.find("dynamics")
        if de is not None:
            j.damping = float(de.get("damping", "0"))
            j.friction = float(de.get("friction", "0"))
        j.dont_collapse = je.get("dont_collapse", "false").lower() == "true"
        joints.append(j)
    return links, joints


# ----------------------------------------------------------------------------- compiled model
def _merge_inertial(parts):
    """parts: list of (mass, com[3], I_com[3,3]) in one frame -> merged triple."""
    m = sum(p[0] for p in parts)
    if m <= 0.0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    com = sum(p[0] * p[1] for p in parts) / m
    I = np.zeros((3, 3))
    for mi, ci, Ii in parts:
        d = ci - com
        I += Ii + mi * (d @ d * np.eye(3) - np.outer(d, d))
    return m, com, I


def _box_points(size, R, p, rep, sid):
    """Sphere-swept stand-in for a box (the contact model works on spheres).  A rod-like box (one side >= 2.5x the others,
    e.g. the A1 thigh 0.2 x 0.0245 x 0.034) becomes a capsule: two end spheres on the long axis with the mean half-thickness
    as radius.  Any other box (robot trunks) becomes a rounded slab: four spheres of radius = half the smallest side at the
    inset corners of the two larger sides.  Both touch the ground plane exactly where the box faces would."""
    size = np.asarray(size, dtype=np.float64)
    order = np.argsort(size)            # small, mid, large
    sm, md, lg = size[order]
    pts = []
    if lg >= 2.5 * md:
        rad = 0.25 * (sm + md)
        h = max(0.5 * lg - rad, 0.0)
        ax = R[:, order[2]]
        for sgn in (1.0, -1.0):
            pts.append(CollisionPoint(p + sgn * h * ax, rad, rep, sid))
    else:
        rad = 0.5 * sm
        a1, a2 = R[:, order[2]], R[:, order[1]]
        h1, h2 = max(0.5 * lg - rad, 0.0), max(0.5 * md - rad, 0.0)
        for s1 in (1.0, -1.0):
            for s2 in (1.0, -1.0):
                pts.append(CollisionPoint(p + s1 * h1 * a1 + s2 * h2 * a2, rad, rep, sid))
    return pts


@dataclass
class CollisionPoint:
    """A sphere of ``radius`` at ``pos`` (dynamic-body frame).  A capsule is two."""
    pos: np.ndarray
    radius: float
    report_body: int
    shape_id: int


@dataclass
class RobotModel:
    name: str
    body_names: List[str]                 # report bodies, Isaac-Gym order
    dof_names: List[str]
    num_limbs: int
    chain_len: int
    base_mass: float
    base_com: np.ndarray
    base_inertia: np.ndarray              # 3x3 at COM, base axes
    joint_pos: np.ndarray                 # [K*L,3] joint origin in parent dynamic body frame
    joint_rot: np.ndarray                 # [K*L,3,3] joint frame -> parent frame at q=0
    joint_axis: np.ndarray                # [K*L,3] unit axis, joint frame
    body_mass: np.ndarray                 # [K*L] dynamic child body (fixed descendants welded in)
    body_com: np.ndarray                  # [K*L,3]
    body_inertia: np.ndarray              # [K*L,3,3]
    dof_lower: np.ndarray
    dof_upper: np.ndarray
    dof_has_limits: np.ndarray
    dof_effort: np.ndarray
    dof_velocity: np.ndarray
    dof_damping: np.ndarray
    dof_friction: np.ndarray
    base_points: List[CollisionPoint]
    limb_points: List[List[CollisionPoint]]   # per limb; .joint attr via point_joint
    limb_point_joint: List[List[int]]         # per limb, per point: index in chain (0..L-1) of carrying body
    report_parent_dyn: List[int]              # per report body: dynamic body id (0 base, 1+limb*L+j)
    report_offset: np.ndarray                 # [nb,3] report-body origin in its dynamic body frame
    report_rot: np.ndarray                    # [nb,3,3]
    report_mass: np.ndarray                   # [nb] mass of each report body alone (known-answer tests)
    num_shapes: int

    @property
    def num_dof(self) -> int:
        return self.num_limbs * self.chain_len

    @property
    def num_bodies(self) -> int:
        return len(self.body_names)

    @property
    def total_mass(self) -> float:
        return float(self.base_mass + self.body_mass.sum())

    # -- name lookups the env does by substring (legged_robot.py:696-702) -------
    def bodies_matching(self, substr: str) -> List[int]:
        return [i for i, n in enumerate(self.body_names) if substr in n]

    # -- forward kinematics (float64, for tests and fixtures) -------------------
    def forward_kinematics(self, q):
        """Returns (R[K*L,3,3], p[K*L,3]) of every dynamic child body in the base frame."""
        q = np.asarray(q, dtype=np.float64)
        K, L = self.num_limbs, self.chain_len
        Rs, ps = np.zeros((K * L, 3, 3)), np.zeros((K * L, 3))
        for k in range(K):
            R, p = np.eye(3), np.zeros(3)
            for j in range(L):
                i = k * L + j
                p = p + R @ self.joint_pos[i]
                R = R @ self.joint_rot[i] @ axis_angle_matrix(self.joint_axis[i], q[i])
                Rs[i], ps[i] = R, p
        return Rs, ps

    def report_body_positions(self, q):
        Rs, ps = self.forward_kinematics(q)
        out = np.zeros((self.num_bodies, 3))
        for b, d in enumerate(self.report_parent_dyn):
            out[b] = self.report_offset[b] if d == 0 else ps[d - 1] + Rs[d - 1] @ self.report_offset[b]
        return out

    def center_of_mass(self, q):
        Rs, ps = self.forward_kinematics(q)
        acc = self.base_mass * self.base_com
        for i in range(self.num_dof):
            acc = acc + self.body_mass[i] * (ps[i] + Rs[i] @ self.body_com[i])
        return acc / self.total_mass

    # -- serialisation ------------------------------------------------------------
    def to_json(self) -> str:
        def enc(o):
            if isinstance(o, np.ndarray):
                return o.tolist()
            if isinstance(o, CollisionPoint):
                return {"pos": o.pos.tolist(), "radius": o.radius, "report_body": o.report_body, "shape_id": o.shape_id}
            if isinstance(o, (np.floating, np.integer, np.bool_)):
                return o.item()
            raise TypeError(type(o))
        return json.dumps(self.__dict__, default=enc, indent=1)

    @staticmethod
    def from_json(text: str) -> "RobotModel":
        d = json.loads(text)
        arr = lambda k: np.asarray(d[k], dtype=np.float64)
        cp = lambda e: CollisionPoint(np.asarray(e["pos"], dtype=np.float64), float(e["radius"]),
                                      int(e["report_body"]), int(e["shape_id"]))
        return RobotModel(
            name=d["name"], body_names=list(d["body_names"]), dof_names=list(d["dof_names"]),
            num_limbs=int(d["num_limbs"]), chain_len=int(d["chain_len"]),
            base_mass=float(d["base_mass"]), base_com=arr("base_com"), base_inertia=arr("base_inertia"),
            joint_pos=arr("joint_pos"), joint_rot=arr("joint_rot"), joint_axis=arr("joint_axis"),
            body_mass=arr("body_mass"), body_com=arr("body_com"), body_inertia=arr("body_inertia"),
            dof_lower=arr("dof_lower"), dof_upper=arr("dof_upper"),
            dof_has_limits=np.asarray(d["dof_has_limits"], dtype=bool),
            dof_effort=arr("dof_effort"), dof_velocity=arr("dof_velocity"),
            dof_damping=arr("dof_damping"), dof_friction=arr("dof_friction"),
            base_points=[cp(e) for e in d["base_points"]],
            limb_points=[[cp(e) for e in lp] for lp in d["limb_points"]],
            limb_point_joint=[list(map(int, lp)) for lp in d["limb_point_joint"]],
            report_parent_dyn=list(map(int, d["report_parent_dyn"])),
            report_offset=arr("report_offset"), report_rot=arr("report_rot"),
            report_mass=arr("report_mass"), num_shapes=int(d["num_shapes"]))


def compile_urdf(path: str, collapse_fixed_joints=True, replace_cylinder_with_capsule=True,
                 name: Optional[str] = None) -> RobotModel:
    links, joints = _parse_urdf(path)
    children: Dict[str, List[_Joint]] = {n: [] for n in links}
    child_names = set()
    for j in joints:
        children[j.parent].append(j)
        child_names.add(j.child)
    roots = [n for n in links if n not in child_names]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")
    root = roots[0]
    for n in children:
        children[n].sort(key=lambda j: j.child)     # alphabetical siblings [EXTERNAL order]

    # A "report body" = root, child of a non-fixed joint, or child of a dont_collapse fixed joint.
    # A "dynamic body" = root or child of a non-fixed joint.
    report_names: List[str] = []
    report_parent_dyn: List[int] = []
    report_T: List[tuple] = []          # (R,p) report frame -> its dynamic body frame
    report_mass: List[float] = []
    dyn = []                            # dicts: name, joint, parent_dyn, R,p (joint frame in parent dyn frame), parts, shapes
    shape_counter = [0]

    def add_report(name, dyn_id, R, p):
        report_names.append(name)
        report_parent_dyn.append(dyn_id)
        report_T.append((R.copy(), p.copy()))
        report_mass.append(0.0)
        return len(report_names) - 1

    def absorb(link_name, dyn_id, rep_id, R, p):
        """Fold `link_name` (frame = R,p in dynamic body `dyn_id`) and its fixed subtree in."""
        lk = links[link_name]
        d = dyn[dyn_id]
        if lk.mass > 0.0:
            d["parts"].append((lk.mass, R @ lk.com + p, R @ lk.inertia @ R.T))
            report_mass[rep_id] += lk.mass
        for s in lk.shapes:
            d["shapes"].append((s, R @ s.R, R @ s.p + p, rep_id, shape_counter[0]))
            shape_counter[0] += 1
        for j in children[link_name]:
            Rc, pc = R @ j.R, R @ j.p + p
            if j.kind == "fixed":
                if collapse_fixed_joints and not j.dont_collapse:
                    absorb(j.child, dyn_id, rep_id, Rc, pc)
                else:
                    rid = add_report(j.child, dyn_id, Rc, pc)
                    absorb(j.child, dyn_id, rid, Rc, pc)
            elif j.kind in ("revolute", "continuous"):
                dyn.append({"name": j.child, "joint": j, "parent": dyn_id, "R": Rc, "p": pc, "parts": [], "shapes": []})
                new_id = len(dyn) - 1
                rid = add_report(j.child, new_id, np.eye(3), np.zeros(3))
                absorb(j.child, new_id, rid, np.eye(3), np.zeros(3))
            else:
                raise ValueError(f"joint type {j.kind!r} of {j.name} is not supported on this path")

    dyn.append({"name": root, "joint": None, "parent": -1, "R": np.eye(3), "p": np.zeros(3), "parts": [], "shapes": []})
    rid0 = add_report(root, 0, np.eye(3), np.zeros(3))
    absorb(root, 0, rid0, np.eye(3), np.zeros(3))

    # ---- limb decomposition: every non-root dynamic body lies on a serial chain off the base
    kids: Dict[int, List[int]] = {i: [] for i in range(len(dyn))}
    for i, d in enumerate(dyn[1:], start=1):
        kids[d["parent"]].append(i)
    chains: List[List[int]] = []
    for first in kids[0]:
        chain, cur = [first], first
        while kids[cur]:
            if len(kids[cur]) != 1:
                raise ValueError("limbs must be serial chains (branching below the base is out of scope)")
            cur = kids[cur][0]
            chain.append(cur)
        chains.append(chain)
    K = len(chains)
    L = len(chains[0]) if K else 0
    if any(len(c) != L for c in chains):
        raise ValueError("all limbs must have the same number of joints")
    # depth-first creation order already makes dyn ids 1+k*L+j; assert it
    for k, c in enumerate(chains):
        assert c == list(range(1 + k * L, 1 + (k + 1) * L)), "unexpected body ordering"

    n = K * L
    m0, c0, I0 = _merge_inertial(dyn[0]["parts"])
    jp, jr, ja = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3))
    bm, bc, bi = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3, 3))
    lo, hi, hl, ef, ve, da, fr = (np.zeros(n) for _ in range(7))
    dof_names = []
    for i in range(n):
        d = dyn[1 + i]
        j: _Joint = d["joint"]
        jp[i], jr[i] = d["p"], d["R"]
        ja[i] = j.axis / np.linalg.norm(j.axis)
        bm[i], bc[i], bi[i] = _merge_inertial(d["parts"])
        lo[i], hi[i], hl[i] = j.lower, j.upper, float(j.has_limits)
        ef[i], ve[i], da[i], fr[i] = j.effort, j.velocity, j.damping, j.friction
        dof_names.append(j.name)

    def points_of(dyn_id):
        pts = []
        for s, R, p, rep, sid in dyn[dyn_id]["shapes"]:
            kind = s.kind
            if kind == "cylinder":
                if not replace_cylinder_with_capsule:
                    raise ValueError("plain cylinders are not supported; use replace_cylinder_with_capsule")
                kind = "capsule"
            if kind == "sphere":
                pts.append(CollisionPoint(p.copy(), s.radius, rep, sid))
            elif kind == "capsule":
                h = 0.5 * s.length
                ax = R[:, 2]
                pts.append(CollisionPoint(p + h * ax, s.radius, rep, sid))
                pts.append(CollisionPoint(p - h * ax, s.radius, rep, sid))
            elif kind == "box":
                pts.extend(_box_points(s.size, R, p, rep, sid))
            else:
                raise ValueError(f"collision primitive {kind!r} is not supported on this path")
        return pts

    base_points = points_of(0)
    limb_points, limb_point_joint = [], []
    for k in range(K):
        pts, pj = [], []
        for jx in range(L):
            for cpnt in points_of(1 + k * L + jx):
                pts.append(cpnt)
                pj.append(jx)
        limb_points.append(pts)
        limb_point_joint.append(pj)

    return RobotModel(
        name=name or os.path.splitext(os.path.basename(path))[0],
        body_names=report_names, dof_names=dof_names, num_limbs=K, chain_len=L,
        base_mass=m0, base_com=c0, base_inertia=I0,
        joint_pos=jp, joint_rot=jr, joint_axis=ja, body_mass=bm, body_com=bc, body_inertia=bi,
        dof_lower=lo, dof_upper=hi, dof_has_limits=hl.astype(bool), dof_effort=ef, dof_velocity=ve,
        dof_damping=da, dof_friction=fr,
        base_points=base_points, limb_points=limb_points, limb_point_joint=limb_point_joint,
        report_parent_dyn=report_parent_dyn,
        report_offset=np.array([t[1] for t in report_T]), report_rot=np.array([t[0] for t in report_T]),
        report_mass=np.array(report_mass), num_shapes=shape_counter[0])


# ----------------------------------------------------------------------------- shipped tables
_MODEL_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), "resources", "models")


def load_model(asset_file: str, **opts) -> RobotModel:
    """Resolve ``cfg.asset.file``: a URDF on disk is compiled; otherwise the
    compiled table ``resources/models/<stem>.json`` (generated by
    ``tools/compile_models.py`` from the URDF of the same stem) is loaded."""
    if asset_file and os.path.isfile(asset_file) and asset_file.endswith(".urdf"):
        return compile_urdf(asset_file, **opts)
    stem = os.path.splitext(os.path.basename(asset_file))[0]
    path = os.path.join(_MODEL_DIR, stem + ".json")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"neither URDF {asset_file!r} nor compiled model {path!r} exists")
    with open(path) as fh:
        return RobotModel.from_json(fh.read())
