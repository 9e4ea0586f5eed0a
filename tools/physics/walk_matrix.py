"""Walk-policy sensitivity matrix on the CPU oracle (test infrastructure): payload x friction, fraction of envs fallen after the
settle window, tracking error.  Physics variants are selected with WALK_SOLVER / WALK_FRICTION / WALK_ITERS / WALK_ERP (cfg.sim.physx overrides, see walk_diag.py)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
from walk_diag import run

def matrix(n=96, steps=400, label=""):
    rows = []
    t0 = time.time()
    for payload in (-5.0, 0.0, 5.0):
        for mu in (0.5, 1.0, 1.5):
            r = run(n=n, steps=steps, payload=payload, mu=mu, verbose=False)
            rows.append((payload, mu, r))
    print(f"== {label}  ({time.time() - t0:.0f} s)")
    print("payload   mu   fallen early  track_err  duty                      base_z  gx")
    for payload, mu, r in rows:
        print(f"{payload:+5.0f}   {mu:4.1f}   {r['frac_fallen']:5.3f}  {r['early_fallen']:5.3f}  {r['track_err']:.3f}    {r['duty']}  {r['base_z']:.3f}  {r['pitch_gx']:+.3f}")
    return rows

if __name__ == "__main__":
    matrix(label=" ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("WALK_")) or "defaults")
