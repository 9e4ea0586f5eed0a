"""Rollout collection, serial against pipelined (review item: two half-batch contexts on two streams, policy(A) next to physics(B)).  24 steps of
act -> env.step on the headline workload: (a) one context of 4096 envs on one stream, (b) two contexts of 2048 envs, each with its own stream, their
chains enqueued side by side.  Also the single kernels at half size, which is what decides the outcome: a physics launch of 2048 envs and a policy launch
of 2048 rows take as long as the full-size ones (one round of workgroups whose length is a latency), so two half chains side by side take what one full chain
takes.  One JSON line."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from extended_legged_gym_amd.rl import NativeActorCritic  # noqa: E402
from tools.bench_rollout import torch_net  # noqa: E402


def main():
    A, T = 12, 24
    torch.manual_seed(0)
    actor, critic = torch_net([235, 512, 256, 128, A]), torch_net([235, 512, 256, 128, 1])
    sd = {"actor." + k: v for k, v in actor.state_dict().items()}
    sd.update({"critic." + k: v for k, v in critic.state_dict().items()})
    sd["std"] = torch.ones(A, device="cuda")
    acs = [NativeActorCritic(sd, "elu", device="cuda:0", seed=1 + i) for i in range(3)]
    out = {}

    def timeit(fn, warm=3, steps=10):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    for n in (4096, 2048, 1024):
        obs = torch.randn(n, 235, device="cuda")
        out[f"policy_act_{n}_rows_ms"] = timeit(lambda: acs[0].act_and_evaluate(obs), 20, 200) * 1e3
    env, _ = bench.build_env(0, 1, 4096, False)
    env.reset()
    o = [env.get_observations()]

    def serial():
        for _ in range(T):
            o[0], _, _, _, _ = env.step(acs[0].act_and_evaluate(o[0])[0])
    a4 = torch.randn(4096, A, device="cuda")
    out["physics_step_4096_ms"] = timeit(lambda: env.step(a4), 50, 300) * 1e3
    out["collect_24_steps_serial_4096_ms"] = timeit(serial) * 1e3
    env.core.close()
    halves = []
    for i in range(2):
        e, _ = bench.build_env(0, 1, 2048, False)
        e.reset()
        halves.append([e, e.get_observations(), torch.cuda.Stream(), acs[1 + i]])
    a2 = torch.randn(2048, A, device="cuda")
    out["physics_step_2048_ms"] = timeit(lambda: halves[0][0].step(a2), 50, 300) * 1e3
    torch.cuda.synchronize()

    def pipelined():
        for _ in range(T):
            for h in halves:
                with torch.cuda.stream(h[2]):
                    h[1], _, _, _, _ = h[0].step(h[3].act_and_evaluate(h[1])[0])
    out["collect_24_steps_two_streams_2x2048_ms"] = timeit(pipelined) * 1e3

    def one_half():
        h = halves[0]
        with torch.cuda.stream(h[2]):
            for _ in range(T):
                h[1], _, _, _, _ = h[0].step(h[3].act_and_evaluate(h[1])[0])
    out["collect_24_steps_one_half_2048_ms"] = timeit(one_half) * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
