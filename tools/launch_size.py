"""Development aid: time of ONE launch of the stage kernel against the number of elements it covers (the same mesh, the same
work per element): the small-launch efficiency that bounds strong scaling before any byte crosses xGMI.

    python tools/launch_size.py [--order 3 --rs 5 --lo 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper

ap = argparse.ArgumentParser()
ap.add_argument("--order", type=int, default=3)
ap.add_argument("--rs", type=int, default=5)
ap.add_argument("--lo", type=int, default=5)
args = ap.parse_args()
lib = bind_driver(load_library())
case = Case(lib, make_config("periodic-cube", args.rs, args.order, 10, -1.0, 0.5, lo_type=args.lo, pa=1))
st = Stepper(lib, case, device="cuda:0")
for _ in range(3):
    st.step(case.dt)
c, u, dt = st.ctx, st.x, case.dt
y = torch.empty_like(u)
ne, nd = case.ne_owned, case.ndof
c.setup(st.t)
c.enable_timers(True)
sizes = [n**3 for n in (12, 16, 24, 32, 36, 48, 64, 72, 96) if n**3 <= ne]
if ne not in sizes:
    sizes.append(ne)
full = None
rows = []
for n in reversed(sizes):
    reps = max(20, min(400, int(2e7 / n)))
    for _ in range(5):
        c.stage_fused_range(u, dt, y, 0, n, True)
    torch.cuda.synchronize()
    c.reset_timers()
    for _ in range(reps):
        c.stage_fused_range(u, dt, y, 0, n, True)
    torch.cuda.synchronize()
    ms = 1e3 * c.timers()[0] / reps
    rate = 1e-6 * n * nd / (1e-3 * ms)
    full = full or rate
    rows.append((n, ms, rate, rate / full))
print(f"p = {args.order}, -rs {args.rs} mesh, lo {args.lo}: one launch over the first n elements (HIP events around the kernel)")
print(f"{'elements':>10s} {'n^(1/3)':>8s} {'ms':>9s} {'MDOFs*stage/s':>14s} {'of full':>8s}")
for n, ms, rate, rel in reversed(rows):
    print(f"{n:10d} {round(n ** (1 / 3)):8d} {ms:9.4f} {rate:14.1f} {rel:8.3f}")
