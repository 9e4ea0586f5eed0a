"""Reduce the reference's robot URDFs to the JSON robot models shipped in the package.

BUILD-CONTAINER ONLY (reads /root/reference/legged_gym/resources/robots).  The JSON is data: masses, frames,
collision spheres and limits after the fixed-joint collapse of `extended_legged_gym_amd/utils/urdf.py`.
Also exports the ANYdrive LSTM actuator weights (`resources/actuator_nets/anydrive_v3_lstm.pt`, 969 floats).
"""
import os, sys, json
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from extended_legged_gym_amd.utils.urdf import load_urdf, save_model

REF = "/root/reference/legged_gym/resources"
OUT = os.path.join(REPO, "extended_legged_gym_amd", "resources")

m = load_urdf(f"{REF}/robots/anymal_c/urdf/anymal_c.urdf", "FOOT", ["SHANK", "THIGH"], ["base"])
save_model(m, f"{OUT}/robots/anymal_c.json")
print("anymal_c", m["num_bodies"], m["body_names"], "mass", m["base_mass"] + sum(map(sum, m["link_mass"])), m["cp_count"])
m = load_urdf(f"{REF}/robots/anymal_b/urdf/anymal_b.urdf", "FOOT", ["SHANK", "THIGH"], ["base"])
save_model(m, f"{OUT}/robots/anymal_b.json")
print("anymal_b", m["num_bodies"], "mass", m["base_mass"] + sum(map(sum, m["link_mass"])), m["cp_count"])
m = load_urdf(f"{REF}/robots/a1/urdf/a1.urdf", "foot", ["thigh", "calf"], ["base"])
save_model(m, f"{OUT}/robots/a1.json")
print("a1", m["num_bodies"], m["body_names"], "mass", m["base_mass"] + sum(map(sum, m["link_mass"])), m["cp_count"])

m = load_urdf(f"{REF}/robots/go2/urdf/go2_description.urdf", "foot", ["thigh", "calf"], ["base", "Head_upper"])
save_model(m, f"{OUT}/robots/go2_description.json")
print("go2", m["num_bodies"], m["body_names"], "mass", m["base_mass"] + sum(map(sum, m["link_mass"])), m["cp_count"])

# ElSpider Air (envs/elspider_air/mixed_terrains/elspider_air_rough_config.py:110-117): six legs; the URDF's collision geometry is STL meshes
# (convex hulls in PhysX) -- the box / sphere approximation the reference ships as el_mini_collsp.urdf stands in for them, the foot keeps
# el_mini.urdf's own 2 cm sphere
m = load_urdf(f"{REF}/robots/el_mini/urdf/el_mini.urdf", "FOOT", ["THIGH", "HIP"], ["trunk"], collision_urdf=f"{REF}/robots/el_mini/urdf/el_mini_collsp.urdf")
save_model(m, f"{OUT}/robots/el_mini.json")
print("el_mini", m["num_bodies"], m["body_names"], "mass", m["base_mass"] + sum(map(sum, m["link_mass"])), m["cp_count"], m["dof_names"])

# the main-rollout tasks of the hexapod name the collision-sphere URDF itself (elspider_air_batch_rollout_config.py:196)
m = load_urdf(f"{REF}/robots/el_mini/urdf/el_mini_collsp.urdf", "FOOT", ["base", "HIP", "THIGH", "SHANK"], [])
save_model(m, f"{OUT}/robots/el_mini_collsp.json")
print("el_mini_collsp", m["num_bodies"], "mass", m["base_mass"] + sum(map(sum, m["link_mass"])), m["cp_count"])

# Cassie (envs/cassie/cassie_config.py:76-82): two legs of six revolute joints, an open chain (cassie.urdf:343-349, 397-403: the knee-spring joints are comments)
m = load_urdf(f"{REF}/robots/cassie/urdf/cassie.urdf", "toe", [], ["pelvis"])
save_model(m, f"{OUT}/robots/cassie.json")
print("cassie", m["num_bodies"], m["body_names"], "mass", m["base_mass"] + sum(map(sum, m["link_mass"])), m["cp_count"], m["dof_names"])

import torch
net = torch.jit.load(f"{REF}/actuator_nets/anydrive_v3_lstm.pt")
sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
order = ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
         "lstm.weight_ih_l1", "lstm.weight_hh_l1", "lstm.bias_ih_l1", "lstm.bias_hh_l1", "linear.weight", "linear.bias"]
flat = np.concatenate([sd[k].reshape(-1) for k in order]).astype(np.float32)
assert flat.size == 969
json.dump(dict(params=[float(x) for x in flat], in_scale=[float(x) for x in sd["in_scale"].reshape(-1)],
               out_scale=float(sd["out_scale"].reshape(-1)[0]), layout=order),
          open(f"{OUT}/actuator_nets/anydrive_v3_lstm.json", "w"))
print("actuator net exported", flat.size)
