"""URDF → reduced legged-robot model (what `gym.load_asset` does for the reference).

Replaces the Isaac Gym URDF importer behind `legged_robot.py:738-763` for the robots the hot path
supports: a floating base with four or six 3-revolute-joint legs, or two 6-revolute-joint legs (the kernel instances of `csrc/lg_instance.h`).  Follows the asset options the reference sets
(`legged_robot_config.py:159-180`): `collapse_fixed_joints=True` (links behind fixed joints are merged into
their parent: inertia, collision shapes, child joints), `dont_collapse="true"` fixed joints keep their child
as a reported rigid body (the FOOT links, `anymal_c.urdf:701`), `replace_cylinder_with_capsule=True`.
Bodies and DOFs are ordered depth-first with children sorted by name, which reproduces Isaac Gym's
LF, LH, RF, RH order for ANYmal-C (`anymal_c_rough_config.py:43-58`).

Collision shapes are reduced to spheres: sphere → itself; capsule → spheres at both segment ends (+ centre when
long), each with the segment to the next sphere of the capsule (`cp_slide`, `include/lgstep.h`: the slot also holds the contact of that piece of the
capsule with a terrain EDGE, which would pass between two spheres); box → its 8 corners.  Mesh collisions (PhysX cooks a convex hull per mesh) cannot be reduced from the URDF alone: a link whose
collision geometry is a mesh, or which has none, takes the primitives of the same-named link of `collision_urdf` when one is given --
`el_mini_collsp.urdf`, the box / sphere approximation of the hexapod's STL hulls that ships next to `el_mini.urdf` -- and is
ignored otherwise.
"""
import json
import xml.etree.ElementTree as ET

import numpy as np

MAX_CP = 8
LEG_COUNTS = (4, 6, 2)
JOINTS_PER_LEG = {4: 3, 6: 3, 2: 6}
MAX_SC_PAIRS = 96


def rpy_to_mat(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _vec(s, n=3, default=0.0):
    if s is None:
        return np.full(n, default)
    return np.array([float(x) for x in s.split()])


def _origin(el):
    o = el.find("origin") if el is not None else None
    if o is None:
        return np.eye(3), np.zeros(3)
    return rpy_to_mat(_vec(o.get("rpy"))), _vec(o.get("xyz"))


def _shift(d):
    return np.dot(d, d) * np.eye(3) - np.outer(d, d)


class _Body:
    def __init__(self, name):
        self.name = name
        self.mass = 0.0
        self.com = np.zeros(3)
        self.inertia = np.zeros((3, 3))   # about com, body axes
        self.spheres = []                 # (pos, radius, slide half-vector)
        self.spheres_compact = []         # the reduced set, used when a leg overflows the contact slots
        self.children = []                # (joint dict, _Body)
        self.slide = True                 # capsule parts of this body slide (False: the trunk)

    def add_inertia(self, m, c, I):
        if m <= 0.0:
            return
        mt = self.mass + m
        cn = (self.mass * self.com + m * c) / mt
        self.inertia = self.inertia + self.mass * _shift(self.com - cn) + I + m * _shift(c - cn)
        self.mass, self.com = mt, cn


def _capsule_parts(p, axis_half, r, parts, slide=True):
    """The spheres of a capsule with axis [p - axis_half, p + axis_half]: its two ends (+ the middle for three), each with the vector to the NEXT
    sphere of the chain (`cp_slide`: the segment of the capsule's axis whose edge contacts the slot also holds, `include/lgstep.h`; zero for the
    last sphere).  A chain whose spheres overlap (d <= 2 r: an edge cannot reach the axis between them; ANYmal's drive housings) carries none, and
    neither do the trunk's capsules (`slide=False`: a trunk contact ends the episode, where on the capsule it is found first matters little)."""
    z = np.zeros(3)
    step = 2.0 * axis_half / (parts - 1)
    d = float(np.linalg.norm(step))
    seg = step if (slide and d > 2.0 * r) else z
    if parts == 3:
        return [(p + axis_half, r, z), (p - axis_half, r, seg), (p, r, seg)]
    return [(p + axis_half, r, z), (p - axis_half, r, seg)]


def _link_spheres(link, compact=False, slide=True):
    """Collision primitives of one URDF link as spheres (centre in the link frame, radius, slide half-vector): sphere -> itself; cylinder /
    capsule -> two parts (three when long), see `_capsule_parts`; box -> its 8 corners as points.

    `compact=True` is the reduced set used when a leg does not fit the MAX_CP contact slots: a slender box (longest edge
    >= 4x the others) becomes a capsule along that edge (two parts, radius = the mean half-width), a stubby cylinder
    (length < 2 r) one sphere, and long capsules get two parts instead of three."""
    out = []
    z = np.zeros(3)
    for col in link.findall("collision"):
        R, p = _origin(col)
        g = col.find("geometry")
        if g is None or len(g) == 0:
            continue
        s = g[0]
        if s.tag == "sphere":
            out.append((p, float(s.get("radius")), z))
        elif s.tag in ("cylinder", "capsule"):
            L, r = float(s.get("length")), float(s.get("radius"))
            if compact and L < 2.0 * r:
                out.append((p, r, z))
                continue
            out += _capsule_parts(p, R @ np.array([0, 0, 0.5 * L]), r, 3 if (L > 4.0 * r and not compact) else 2, slide)
        elif s.tag == "box":
            sz = _vec(s.get("size"))
            ax = int(np.argmax(sz))
            others = [sz[i] for i in range(3) if i != ax]
            if compact and sz[ax] >= 4.0 * max(others):
                r = 0.25 * (others[0] + others[1])
                d = np.zeros(3); d[ax] = 0.5 * sz[ax] - r
                out += _capsule_parts(p, R @ d, r, 2, slide)
                continue
            for sx in (-0.5, 0.5):
                for sy in (-0.5, 0.5):
                    for szz in (-0.5, 0.5):
                        out.append((p + R @ (sz * np.array([sx, sy, szz])), 0.0, z))
    return out


def _two_extremes(spheres):
    """At most two primitives of a link in the reduced set: the pair that is farthest apart (its two ends)."""
    if len(spheres) <= 2:
        return list(spheres)
    best, pair = -1.0, (0, 1)
    for i in range(len(spheres)):
        for j in range(i + 1, len(spheres)):
            d = float(np.linalg.norm(np.asarray(spheres[i][0]) - np.asarray(spheres[j][0])))      # (pos, radius, slide)
            if d > best:
                best, pair = d, (i, j)
    return [spheres[pair[0]], spheres[pair[1]]]


def _has_primitive_collision(link):
    for col in link.findall("collision"):
        g = col.find("geometry")
        if g is not None and len(g) and g[0].tag in ("sphere", "cylinder", "capsule", "box"):
            return True
    return False


def load_urdf(path, foot_name, penalize_contacts_on, terminate_after_contacts_on, collapse_fixed_joints=True, collision_urdf=None):
    """Returns a plain-dict robot model with the fields of `lg_robot_model` (include/lgstep.h)."""
    root = ET.parse(path).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    col_links = {}
    if collision_urdf is not None:
        col_links = {l.get("name"): l for l in ET.parse(collision_urdf).getroot().findall("link") if _has_primitive_collision(l)}

    def collision_source(name):
        """The element whose <collision> primitives stand for link `name` (see the module docstring)."""
        return links[name] if _has_primitive_collision(links[name]) or name not in col_links else col_links[name]
    joints = []
    child_names = set()
    for j in root.findall("joint"):
        R, p = _origin(j)
        lim = j.find("limit")
        ax = j.find("axis")
        joints.append(dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
                           child=j.find("child").get("link"), R=R, p=p,
                           axis=_vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0, 0]),
                           lower=float(lim.get("lower")) if lim is not None and lim.get("lower") else 0.0,
                           upper=float(lim.get("upper")) if lim is not None and lim.get("upper") else 0.0,
                           effort=float(lim.get("effort", 0)) if lim is not None else 0.0,
                           velocity=float(lim.get("velocity", 0)) if lim is not None else 0.0,
                           keep=j.get("dont_collapse") == "true"))
        child_names.add(j.find("child").get("link"))
    roots = [n for n in links if n not in child_names]
    assert len(roots) == 1, f"expected a single root link, got {roots}"
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j["parent"], []).append(j)

    def absorb(body, link_name, R, p):
        """Merge URDF link `link_name`, placed at (R, p) in `body`'s frame, and everything fixed to it."""
        link = links[link_name]
        ine = link.find("inertial")
        if ine is not None:
            Ri, pi = _origin(ine)
            m = float(ine.find("mass").get("value"))
            t = ine.find("inertia")
            I = np.array([[float(t.get("ixx")), float(t.get("ixy")), float(t.get("ixz"))],
                          [float(t.get("ixy")), float(t.get("iyy")), float(t.get("iyz"))],
                          [float(t.get("ixz")), float(t.get("iyz")), float(t.get("izz"))]])
            Rw = R @ Ri
            body.add_inertia(m, p + R @ pi, Rw @ I @ Rw.T)
        for (sp, sr, ss) in _link_spheres(collision_source(link_name), slide=body.slide):
            body.spheres.append((p + R @ sp, sr, R @ ss))
        for (sp, sr, ss) in _link_spheres(collision_source(link_name), compact=True, slide=body.slide):
            body.spheres_compact.append((p + R @ sp, sr, R @ ss))
        for j in sorted(by_parent.get(link_name, []), key=lambda jj: jj["child"]):
            Rj, pj = R @ j["R"], p + R @ j["p"]
            if j["type"] == "fixed" and collapse_fixed_joints and not j["keep"]:
                absorb(body, j["child"], Rj, pj)
            else:
                cb = _Body(j["child"])
                absorb(cb, j["child"], np.eye(3), np.zeros(3))
                body.children.append((dict(j, R=Rj, p=pj), cb))

    base = _Body(roots[0])
    base.slide = False
    absorb(base, roots[0], np.eye(3), np.zeros(3))
    base.children.sort(key=lambda jc: jc[1].name)
    legs = [jc for jc in base.children if jc[0]["type"] in ("revolute", "continuous")]
    assert len(legs) in LEG_COUNTS, f"hot path supports robots with {LEG_COUNTS} legs, found {len(legs)} revolute children of the base"
    num_legs = len(legs)
    nj = JOINTS_PER_LEG[num_legs]

    m = dict(base_mass=base.mass, base_com=base.com.tolist(), base_inertia=_sym6(base.inertia),
             joint_pos=[], joint_rot=[], joint_axis=[], link_mass=[], link_com=[], link_inertia=[],
             foot_pos=[], foot_rot=[], dof_lower=[], dof_upper=[], dof_vel_limit=[], torque_limit=[],
             cp_count=[], cp_link=[], cp_body=[], cp_pos=[], cp_radius=[], cp_slide=[])
    body_names, dof_names = [base.name], []
    has_foot = True
    base_spheres = list(base.spheres)
    for l, (j0, b0) in enumerate(legs):
        chain, j, b = [], j0, b0
        foot = None
        while True:
            chain.append((j, b))
            nxt = [jc for jc in b.children if jc[0]["type"] in ("revolute", "continuous")]
            fixed = [jc for jc in b.children if jc[0]["type"] == "fixed"]
            if not nxt:
                foot = fixed[0] if fixed else None
                break
            assert len(nxt) == 1 and not fixed, "legs must be serial chains"
            j, b = nxt[0]
        assert len(chain) == nj, f"leg {b0.name}: expected {nj} revolute joints, got {len(chain)}"
        jp, jr, ja, lm, lc, li = [], [], [], [], [], []
        cps, cps_c = [], []   # (link, body index, pos, radius, slide): full and reduced primitive sets
        for k, (jj, bb) in enumerate(chain):
            body_index = len(body_names)
            body_names.append(bb.name)
            dof_names.append(jj["name"])
            mass, com, ine = bb.mass, bb.com.copy(), bb.inertia.copy()
            spheres = [(k, body_index, sp, sr, ss) for (sp, sr, ss) in bb.spheres]
            spheres_c = [(k, body_index, sp, sr, ss) for (sp, sr, ss) in _two_extremes(bb.spheres_compact)]
            if k == nj - 1:
                if foot is not None:
                    fj, fb = foot
                    tmp = _Body("tmp")
                    tmp.add_inertia(mass, com, ine)
                    tmp.add_inertia(fb.mass, fj["p"] + fj["R"] @ fb.com, fj["R"] @ fb.inertia @ fj["R"].T)
                    mass, com, ine = tmp.mass, tmp.com, tmp.inertia
                    m["foot_pos"].append(fj["p"].tolist())
                    m["foot_rot"].append(fj["R"].reshape(-1).tolist())
                    foot_index = body_index + 1
                    spheres = [(nj, foot_index, fj["p"] + fj["R"] @ sp, sr, fj["R"] @ ss) for (sp, sr, ss) in fb.spheres] + spheres
                    spheres_c = [(nj, foot_index, fj["p"] + fj["R"] @ sp, sr, fj["R"] @ ss) for (sp, sr, ss) in fb.spheres_compact] + spheres_c
                else:
                    has_foot = False
                    m["foot_pos"].append([0.0, 0.0, 0.0])
                    m["foot_rot"].append(np.eye(3).reshape(-1).tolist())
            cps = spheres + cps   # distal links first: feet lead the Gauss-Seidel order
            cps_c = spheres_c + cps_c
            jp.append(jj["p"].tolist())
            jr.append(jj["R"].reshape(-1).tolist())
            ax = jj["axis"] / np.linalg.norm(jj["axis"])
            ja.append(ax.tolist())
            lm.append(mass)
            lc.append(com.tolist())
            li.append(_sym6(ine))
            m["dof_lower"].append(jj["lower"])
            m["dof_upper"].append(jj["upper"])
            m["dof_vel_limit"].append(jj["velocity"])
            m["torque_limit"].append(jj["effort"])
        if foot is not None:
            body_names.append(foot[1].name)
        base_share = [(-1, 0, sp, sr, ss) for bi, (sp, sr, ss) in enumerate(base_spheres) if bi % num_legs == l]
        if len(cps) + len(base_share) > MAX_CP:
            # does not fit the contact slots: reduced primitives for the leg, and the trunk keeps its share (the task
            # terminates on trunk contact, legged_robot.py:215-221) ahead of the proximal links
            cps = cps_c[:max(0, MAX_CP - len(base_share))]
        cps = (cps + base_share)[:MAX_CP]
        pad = MAX_CP - len(cps)
        m["cp_count"].append(len(cps))
        m["cp_link"].append([c[0] for c in cps] + [0] * pad)
        m["cp_body"].append([c[1] for c in cps] + [0] * pad)
        m["cp_pos"].append([np.asarray(c[2]).tolist() for c in cps] + [[0.0, 0.0, 0.0]] * pad)
        m["cp_radius"].append([c[3] for c in cps] + [0.0] * pad)
        m["cp_slide"].append([np.asarray(c[4]).tolist() for c in cps] + [[0.0, 0.0, 0.0]] * pad)
        for key, val in (("joint_pos", jp), ("joint_rot", jr), ("joint_axis", ja), ("link_mass", lm),
                         ("link_com", lc), ("link_inertia", li)):
            m[key].append(val)

    m["has_foot_body"] = int(has_foot)
    m["num_legs"] = num_legs
    m["num_joints_per_leg"] = nj
    m["num_bodies"] = len(body_names)
    m["body_names"], m["dof_names"] = body_names, dof_names
    m["sc_pairs"] = self_collision_pairs(m)
    finalize_indices(m, foot_name, penalize_contacts_on, terminate_after_contacts_on)
    return m


def sphere_centres(m, q):
    """World positions (base frame = identity) of every collision sphere at joint angles q (S, dof): {(leg, slot): (S, 3)}.  Numpy forward
    kinematics of the reduced model (the kernels' `leg_kinematics`), used on the host only."""
    q = np.atleast_2d(np.asarray(q, np.float64))
    S = q.shape[0]
    nl, nj = m["num_legs"], len(m["joint_pos"][0])
    out = {}

    def rodrigues(a, ang):
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3)[None] + np.sin(ang)[:, None, None] * K[None] + (1 - np.cos(ang))[:, None, None] * (K @ K)[None]
    for l in range(nl):
        R = np.broadcast_to(np.eye(3), (S, 3, 3)).copy()
        O = np.zeros((S, 3))
        frames = [(R.copy(), O.copy())]                    # index 0: the base; 1 + j: link j
        for j in range(nj):
            O = O + R @ np.asarray(m["joint_pos"][l][j])
            R = R @ np.asarray(m["joint_rot"][l][j]).reshape(3, 3) @ rodrigues(np.asarray(m["joint_axis"][l][j]), q[:, nj * l + j])
            frames.append((R.copy(), O.copy()))
        for s in range(m["cp_count"][l]):
            link = m["cp_link"][l][s]
            Rl, Ol = frames[0] if link < 0 else frames[1 + min(link, nj - 1)]
            out[(l, s)] = Ol + Rl @ np.asarray(m["cp_pos"][l][s])
    return out


def self_collision_pairs(m, samples=20000, seed=0, reach=0.02):
    """Candidate sphere pairs of the self-collision pass (`lg_robot_model.sc_pairs`): pairs PhysX would test with `asset.self_collisions = 0`
    -- shapes of different bodies that are not parent and child -- that come within `reach` of each other somewhere in the joint-limit box
    (limits; unlimited joints and joints with more than 2 rad of range: +-1 rad around the middle -- ANYmal's URDF limits none of its joints, and a
    hip swung by 1.5 rad puts pairs in front that no gait comes near while the feet of neighbouring legs drop off the list), found by sampling; the most
    frequent MAX_SC_PAIRS when there are more.  Two zero-radius points (box corners) never meet and are left out.  [leg a, slot a, leg b, slot b]."""
    nl, nj = m["num_legs"], len(m["joint_pos"][0])
    lo, hi = np.asarray(m["dof_lower"], np.float64), np.asarray(m["dof_upper"], np.float64)
    wide = ~(lo < hi) | (hi - lo > 2.0)
    mid = np.where(~(lo < hi), 0.0, 0.5 * (lo + hi))
    lo, hi = np.where(wide, mid - 1.0, lo), np.where(wide, mid + 1.0, hi)
    rng = np.random.RandomState(seed)
    q = lo + (hi - lo) * rng.rand(samples, nl * nj)
    cen = sphere_centres(m, q)
    keys = sorted(cen)
    found = []
    for ia, (la, sa) in enumerate(keys):
        for (lb, sb) in keys[ia + 1:]:
            ka, kb = m["cp_link"][la][sa], m["cp_link"][lb][sb]
            ra, rb = m["cp_radius"][la][sa], m["cp_radius"][lb][sb]
            if ra == 0.0 and rb == 0.0:
                continue
            ka, kb = min(ka, nj - 1), min(kb, nj - 1)          # the foot body is fixed to the last link
            if ka < 0 and kb < 0:
                continue                                      # both on the trunk
            if (ka < 0) != (kb < 0):
                if max(ka, kb) == 0:
                    continue                                  # trunk and a first link: parent and child
            elif la == lb and abs(ka - kb) < 2:
                continue                                      # same link, or parent and child
            d = np.linalg.norm(cen[(la, sa)] - cen[(lb, sb)], axis=1) - ra - rb
            hits = int((d < reach).sum())
            if hits:
                found.append((hits, [la, sa, lb, sb]))
    found.sort(key=lambda t: (-t[0], t[1]))
    return sorted(p for _, p in found[:MAX_SC_PAIRS])


def finalize_indices(m, foot_name, penalize_contacts_on, terminate_after_contacts_on):
    """Body-index lists by substring match, as `legged_robot.py:764-770,801-815`."""
    names = m["body_names"]
    m["feet_indices"] = [i for i, s in enumerate(names) if foot_name in s]
    pen = []
    for n in penalize_contacts_on:
        pen.extend([i for i, s in enumerate(names) if n in s])
    term = []
    for n in terminate_after_contacts_on:
        term.extend([i for i, s in enumerate(names) if n in s])
    m["penalised_contact_indices"], m["termination_contact_indices"] = pen, term
    return m


def _sym6(I):
    return [float(I[0, 0]), float(I[0, 1]), float(I[0, 2]), float(I[1, 1]), float(I[1, 2]), float(I[2, 2])]


def save_model(m, path):
    with open(path, "w") as f:
        json.dump(m, f, indent=1)


def load_model(path):
    with open(path) as f:
        return json.load(f)
