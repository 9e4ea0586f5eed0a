"""Diagnostic: time `PPO.act` (fused actor + critic, 4096 x 235-512-256-128) with the library named by LGSTEP_LIB."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extended_legged_gym_amd.rl import NativeActorCritic  # noqa: E402
from tools.bench_rollout import torch_net  # noqa: E402

N, A = int(os.environ.get("AB_N", 4096)), 12
torch.manual_seed(0)
actor, critic = torch_net([235, 512, 256, 128, A]), torch_net([235, 512, 256, 128, 1])
sd = {"actor." + k: v for k, v in actor.state_dict().items()}
sd.update({"critic." + k: v for k, v in critic.state_dict().items()})
sd["std"] = torch.ones(A, device="cuda")
ac = NativeActorCritic(sd, "elu", device="cuda:0", seed=1)
obs = torch.randn(N, 235, device="cuda")
for _ in range(int(os.environ.get("AB_WARM", 50))):
    ac.act_and_evaluate(obs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(int(os.environ.get("AB_ITERS", 300))):
    ac.act_and_evaluate(obs)
torch.cuda.synchronize()
print("lib %s  N %d  act %.2f us" % (os.path.basename(os.environ.get("LGSTEP_LIB", "liblgstep.so")), N, (time.perf_counter() - t0) / int(os.environ.get("AB_ITERS", 300)) * 1e6))
