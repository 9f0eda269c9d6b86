"""Development aid (not collected by pytest): throughput of the dim = 2 path on the MI355X -- HO kernel + fused limiter (or the
RD solver + fused limiter) -- on refined inline-quad meshes.  The inputs come from the oracle's case builder (test
infrastructure), every stage from librmh.so.

    python tests/perf_2d.py [--order 3 --rs 6 --steps 20 --lo 5]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle.remhos_oracle import Config, Remhos
from tests.test_2d import Backend

ap = argparse.ArgumentParser()
ap.add_argument("--order", type=int, default=3)
ap.add_argument("--rs", type=int, default=6)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--lo", type=int, default=5)
args = ap.parse_args()
bk = Backend(True)
r = Remhos(Config(mesh="inline-quad", rs=args.rs, order=args.order, problem=14, dt=-1.0, t_final=0.5, lo=args.lo))
ctx = bk.context(r, pa=True)
x = bk.arr(r.u)
y, k, dulo = (torch.zeros_like(x) for _ in range(3))
dt = r.dt


def stage(u, t, x_base, a, b, out):
    ctx.setup(t)
    ctx.ho_apply(u, k)
    if args.lo == 5:
        ctx.limit_fused(u, k, dt, x_base=x_base, a=a, b=b, dt_rk=dt, y_out=out)
    else:
        ctx.lo_rdsubcell(u, dulo)
        ctx.limit_fused_lo(u, k, dulo, dt, x_base=x_base, a=a, b=b, dt_rk=dt, y_out=out)


def step(t):
    stage(x, t, None, 0.0, 1.0, y)
    stage(y, t + dt, x, 0.75, 0.25, y)
    stage(y, t + dt / 2, x, 1.0 / 3.0, 2.0 / 3.0, x)


t = 0.0
for _ in range(3):
    step(t)
    t += dt
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step(t)
    t += dt
torch.cuda.synchronize()
el = time.perf_counter() - t0
n = x.numel()
print(f"inline-quad -rs {args.rs} -o {args.order} -lo {args.lo}: {r.lat.ne} elements, {n} dofs, {1e3 * el / args.steps:.3f} ms/step, "
      f"{1e-6 * n * 3 * args.steps / el:.1f} MDOFs*stage/s (cg {ctx.last_cg_iters()})")
