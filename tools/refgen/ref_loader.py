"""Import the reference's pure-PyTorch env layer in the build container.

BUILD-CONTAINER ONLY.  This is test infrastructure used to generate the golden
vectors under `tests/golden/`; nothing here ships to the GPU box and nothing in
the product imports it.  The reference (`/root/reference`) imports the closed
`isaacgym` package and `warp`/`trimesh`/`cv2`/`torchvision` at module top; none
is installed, so this loader

* installs a stub `isaacgym` package whose `torch_utils` / `terrain_utils` are the
  restatements in `extended_legged_gym_amd.utils` and whose `gymapi.acquire_gym()`
  returns a `FakeGym` (tensors are plain torch-CPU tensors, `simulate()` applies a
  scripted state),
* installs `MagicMock` stand-ins for the other absent third-party modules,
* then imports the *real* `legged_gym` package from `/root/reference`.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

# /root/reference must stay untouched: importing from it must not drop __pycache__ directories next to its sources.
sys.dont_write_bytecode = True

REF_ROOT = "/root/reference"
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# ----------------------------------------------------------------------------- gymapi stub
class _NS:
    def __init__(self, *a, **k):
        self.__dict__.update(k)
        self._args = a


class Vec3(_NS):
    def __init__(self, x=0., y=0., z=0.):
        self.x, self.y, self.z = float(x), float(y), float(z)


class Transform(_NS):
    def __init__(self, p=None, r=None):
        self.p = p if p is not None else Vec3()
        self.r = r


class _PhysX(_NS):
    pass


class SimParams(_NS):
    def __init__(self):
        self.dt = 0.005
        self.substeps = 1
        self.use_gpu_pipeline = False
        self.physx = _PhysX()
        self.gravity = Vec3(0, 0, -9.81)


class _Shape:
    def __init__(self):
        self.friction = 1.0


class _Body:
    def __init__(self, mass):
        self.mass = mass


class FakeAsset:
    def __init__(self, dof_names, body_names, dof_props, body_masses):
        self.dof_names = dof_names
        self.body_names = body_names
        self.dof_props = dof_props
        self.body_masses = body_masses


class FakeGym:
    """Minimal stand-in for the Isaac Gym tensor API (`legged_robot.py:97-103,118-120,564-584`)."""
    robot = None   # class-level: dict(dof_names, body_names, lower, upper, velocity, effort, body_masses)

    def __init__(self):
        self.tensors = {}
        self.script = None        # callable(gym) applied on simulate()
        self.num_envs = 0
        self.sim_calls = 0

    # --- sim / terrain creation: no-ops
    def create_sim(self, *a): return "sim"
    def prepare_sim(self, *a): return True
    def add_ground(self, *a): pass
    def add_heightfield(self, *a): pass
    def add_triangle_mesh(self, *a): pass
    def set_light_parameters(self, *a): pass
    def create_viewer(self, *a): return None

    # --- asset
    def load_asset(self, sim, root, file, options):
        r = FakeGym.robot
        props = np.zeros(len(r["dof_names"]), dtype=[("lower", "f4"), ("upper", "f4"), ("velocity", "f4"), ("effort", "f4")])
        props["lower"], props["upper"] = r["lower"], r["upper"]
        props["velocity"], props["effort"] = r["velocity"], r["effort"]
        return FakeAsset(r["dof_names"], r["body_names"], props, r["body_masses"])

    def get_asset_dof_count(self, a): return len(a.dof_names)
    def get_asset_rigid_body_count(self, a): return len(a.body_names)
    def get_asset_dof_properties(self, a): return a.dof_props
    def get_asset_rigid_shape_properties(self, a): return [_Shape() for _ in a.body_names]
    def get_asset_rigid_body_names(self, a): return list(a.body_names)
    def get_asset_dof_names(self, a): return list(a.dof_names)
    def set_asset_rigid_shape_properties(self, a, p): pass

    def create_env(self, *a):
        self.num_envs += 1
        return self.num_envs - 1

    def create_actor(self, env, asset, pose, name, i, sc, z):
        self._asset = asset
        return 0

    def set_actor_dof_properties(self, *a): pass
    def get_actor_rigid_body_properties(self, env, actor): return [_Body(m) for m in self._asset.body_masses]

    def set_actor_rigid_body_properties(self, env, actor, props, recomputeInertia=True):
        self.tensors.setdefault("body_mass", {})[env] = [p.mass for p in props]

    def find_actor_rigid_body_handle(self, env, actor, name): return self._asset.body_names.index(name)

    # --- tensor API
    def _t(self, key, shape):
        if key not in self.tensors:
            self.tensors[key] = torch.zeros(*shape, dtype=torch.float)
        return self.tensors[key]

    def acquire_actor_root_state_tensor(self, sim):
        t = self._t("root", (self.num_envs, 13))
        t[:, 6] = 1.0
        return t

    def acquire_dof_state_tensor(self, sim): return self._t("dof", (self.num_envs * len(self._asset.dof_names), 2))
    def acquire_net_contact_force_tensor(self, sim): return self._t("contact", (self.num_envs * len(self._asset.body_names), 3))
    def acquire_rigid_body_state_tensor(self, sim): return self._t("rigid", (self.num_envs * len(self._asset.body_names), 13))

    def refresh_dof_state_tensor(self, sim): pass
    def refresh_actor_root_state_tensor(self, sim): pass
    def refresh_net_contact_force_tensor(self, sim): pass
    def refresh_rigid_body_state_tensor(self, sim): pass
    def set_dof_actuation_force_tensor(self, sim, t): self.last_torques = t.clone()
    def set_dof_actuation_force_tensor_indexed(self, sim, t, ids, n): self.last_torques = t.clone()
    def set_dof_state_tensor_indexed(self, *a): pass
    def set_actor_root_state_tensor_indexed(self, *a): pass
    def set_actor_root_state_tensor(self, *a): pass
    def set_dof_state_tensor(self, *a): pass
    def fetch_results(self, *a): pass

    def simulate(self, sim):
        if self.script is not None:
            self.script(self, self.sim_calls)
        self.sim_calls += 1


_THE_GYM = [None]


def _acquire_gym():
    _THE_GYM[0] = FakeGym()
    return _THE_GYM[0]


def current_gym():
    return _THE_GYM[0]


def install_stubs():
    if "isaacgym" in sys.modules and getattr(sys.modules["isaacgym"], "_lg_stub", False):
        return
    sys.path.insert(0, REPO_ROOT)
    from extended_legged_gym_amd.utils import isaac_torch_utils as my_tu
    from extended_legged_gym_amd.utils import terrain_utils as my_terr

    isaac = types.ModuleType("isaacgym")
    isaac._lg_stub = True
    isaac.__path__ = []

    gymapi = types.ModuleType("isaacgym.gymapi")
    gymapi.acquire_gym = _acquire_gym
    gymapi.Vec3, gymapi.Transform, gymapi.SimParams = Vec3, Transform, SimParams
    for n in ["AssetOptions", "PlaneParams", "HeightFieldParams", "TriangleMeshParams", "CameraProperties", "Quat"]:
        setattr(gymapi, n, type(n, (_NS,), {"__init__": lambda self, *a, **k: setattr(self, "transform", Transform())}))
    gymapi.SIM_PHYSX, gymapi.SIM_FLEX = 1, 0
    gymapi.KEY_ESCAPE, gymapi.KEY_V = 0, 1
    gymapi.DOF_MODE_EFFORT = 3

    def _gymapi_default(name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.isupper():
            return 0
        return type(name, (_NS,), {})
    gymapi.__getattr__ = _gymapi_default

    gymtorch = types.ModuleType("isaacgym.gymtorch")
    gymtorch.wrap_tensor = lambda t: t
    gymtorch.unwrap_tensor = lambda t: t

    gymutil = types.ModuleType("isaacgym.gymutil")
    gymutil.parse_device_str = lambda s: (s.split(":")[0], int(s.split(":")[1]) if ":" in s else 0)
    gymutil.parse_sim_config = lambda cfg, sp: None
    gymutil.parse_arguments = MagicMock()
    gymutil.WireframeSphereGeometry = MagicMock()
    gymutil.draw_lines = MagicMock()
    gymutil.LineGeometry = type("LineGeometry", (), {})
    gymutil.AxesGeometry = type("AxesGeometry", (gymutil.LineGeometry,), {})

    tu = types.ModuleType("isaacgym.torch_utils")
    for n in dir(my_tu):
        if not n.startswith("_"):
            setattr(tu, n, getattr(my_tu, n))
    terr = types.ModuleType("isaacgym.terrain_utils")
    for n in dir(my_terr):
        if not n.startswith("_"):
            setattr(terr, n, getattr(my_terr, n))

    isaac.gymapi, isaac.gymtorch, isaac.gymutil, isaac.torch_utils, isaac.terrain_utils = gymapi, gymtorch, gymutil, tu, terr
    sys.modules.update({"isaacgym": isaac, "isaacgym.gymapi": gymapi, "isaacgym.gymtorch": gymtorch,
                        "isaacgym.gymutil": gymutil, "isaacgym.torch_utils": tu, "isaacgym.terrain_utils": terr})

    absent = []
    for name in ["warp", "trimesh", "cv2", "torchvision", "git", "tensorboard", "pytorch3d", "traj_sampling",
                 "open3d", "onnx", "wandb", "neptune", "isaac_utils"]:
        try:
            __import__(name)
        except Exception:
            absent.append(name)

    import importlib.abc
    import importlib.machinery

    class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        """Any import below an absent third-party top-level package resolves to a MagicMock module."""
        def find_spec(self, fullname, path, target=None):
            if fullname.split(".")[0] in absent or fullname == "torch.utils.tensorboard":
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            m = MagicMock(name=spec.name)
            m.__path__ = []
            m.__name__ = spec.name
            m.__spec__ = spec
            return m

        def exec_module(self, module):
            pass
    sys.meta_path.insert(0, _MockFinder())

    sys.path.insert(0, os.path.join(REF_ROOT, "rsl_rl"))
    sys.path.insert(0, os.path.join(REF_ROOT, "legged_gym"))


ANYMAL_LEGS = ["LF", "LH", "RF", "RH"]


def anymal_robot_description():
    """DOF / body naming of ANYmal-C after fixed-joint collapse, in Isaac Gym's (alphabetical DFS) order."""
    dof_names, body_names = [], ["base"]
    for leg in ANYMAL_LEGS:
        dof_names += [f"{leg}_HAA", f"{leg}_HFE", f"{leg}_KFE"]
        body_names += [f"{leg}_HIP", f"{leg}_THIGH", f"{leg}_SHANK", f"{leg}_FOOT"]
    n = len(dof_names)
    return dict(dof_names=dof_names, body_names=body_names,
                lower=np.full(n, -9.42, np.float32), upper=np.full(n, 9.42, np.float32),
                velocity=np.full(n, 20.0, np.float32), effort=np.full(n, 80.0, np.float32),
                body_masses=[30.0] + [1.0] * 16)


ELSPIDER_LEGS = ["LB", "LF", "LM", "RB", "RF", "RM"]


def elspider_robot_description():
    """DOF / body naming of ElSpider Air (`el_mini.urdf`) after fixed-joint collapse, in Isaac Gym's (alphabetical DFS) order -- the order
    `ElSpider._reward_gait_2_step` (elspider.py:366) and `AsyncGaitSchedulerCfg.dof_names` (utils/gait_scheduler.py:99-104) spell out --,
    with the URDF's joint limits."""
    dof_names, body_names = [], ["base"]
    for leg in ELSPIDER_LEGS:
        dof_names += [f"{leg}_HAA", f"{leg}_HFE", f"{leg}_KFE"]
        body_names += [f"{leg}_HIP", f"{leg}_THIGH", f"{leg}_SHANK", f"{leg}_FOOT"]
    n = len(dof_names)
    return dict(dof_names=dof_names, body_names=body_names,
                lower=np.tile(np.array([-0.785, -0.5233, -0.6978], np.float32), 6), upper=np.tile(np.array([0.785, 3.14, 3.925], np.float32), 6),
                velocity=np.full(n, 21.0, np.float32), effort=np.full(n, 33.5, np.float32),
                body_masses=[15.8991] + [1.0] * 24)


def cassie_robot_description():
    """DOF / body naming of Cassie (`cassie.urdf`) after fixed-joint collapse, in Isaac Gym's (alphabetical DFS) order, with the URDF's joint limits
    (`cassie.urdf:315-416`)."""
    joints = ["hip_abduction", "hip_rotation", "hip_flexion", "thigh_joint", "ankle_joint", "toe_joint"]
    links = ["pelvis_rotation", "hip", "thigh", "shin", "tarsus", "toe"]
    dof_names = [f"{j}_{side}" for side in ("left", "right") for j in joints]
    body_names = ["pelvis"] + [f"{side}_{l}" for side in ("left", "right") for l in links]
    lower = np.array([-0.2618, -0.3927, -0.8727, -2.8623, 0.6458, -2.4435, -0.3927, -0.3927, -0.8727, -2.8623, 0.6458, -2.4435], np.float32)
    upper = np.array([0.3927, 0.3927, 1.3963, -0.6458, 2.8623, -0.5236, 0.2618, 0.3927, 1.3963, -0.6458, 2.8623, -0.5236], np.float32)
    vel = np.tile(np.array([20.1475, 20.1475, 20.5085, 20.5085, 20.5085, 20.5192], np.float32), 2)
    eff = np.tile(np.array([112.0, 112.0, 195.0, 195.0, 195.0, 45.0], np.float32), 2)
    return dict(dof_names=dof_names, body_names=body_names, lower=lower, upper=upper, velocity=vel, effort=eff, body_masses=[10.33] + [1.0] * 12)


def load_reference():
    """Returns the reference `legged_gym` package (real code) after installing stubs."""
    install_stubs()
    import legged_gym  # noqa: F401  (real package from /root/reference)
    return legged_gym
