import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
env, cfg = bench.build_env(0, 1, 4096, False)
env.reset()
a = torch.randn(4096, 12, device="cuda")
for _ in range(300): env.step(a)
torch.cuda.synchronize()
ts = []
for _ in range(100):
    t0 = time.perf_counter(); env.step(a); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
import numpy as np
ts = np.array(ts) * 1e6
print("host us per env.step: first %.1f, median %.1f, p90 %.1f" % (ts[0], np.median(ts), np.quantile(ts, .9)))
core = env.core
import ctypes as C
st = core._stream(); p = C.c_void_p(a.data_ptr())
ts = []
for _ in range(100):
    t0 = time.perf_counter(); core.lib.lg_step(core.ctx, p, st); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
ts = np.array(ts) * 1e6
print("raw lg_step ctypes call: median %.1f" % np.median(ts))
ts = []
for _ in range(100):
    t0 = time.perf_counter(); core._stream(); ts.append(time.perf_counter() - t0)
print("_stream(): median %.1f us" % (np.median(np.array(ts)) * 1e6))
