"""Development aid: time the GRANULAR entry points of the C ABI -- the reference's call sequence
(ComputeElementsMinMax, ComputeBounds, MassBasedAvg, ClipScale; remhos.cpp:1815-1831) and the fused limiter --
one by one on a real state, with HIP events, and price them against their algorithmic HBM bytes.

    python tools/gbench.py [--order 3 --rs 5 --reps 20] [name ...]      (remhos_amd/librmh_<name>.so; "" / main = librmh.so)

Every kernel here streams E-vectors once: bytes = 8 * ndofs * (vectors read + written) (+ the stencil table for the
bounds).  Results of each library are compared with the first one's (max |difference|).
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
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--mesh", default="periodic-cube")
ap.add_argument("--bt", type=int, default=0)
ap.add_argument("names", nargs="*", default=["main"])
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps  # ms


ref = {}
for name in args.names:
    path = os.path.join(root, "remhos_amd", f"librmh_{name}.so" if name not in ("", "main") else "librmh.so")
    lib = bind_driver(load_library(path))
    case = Case(lib, make_config(args.mesh, args.rs, args.order, 10, -1.0, 0.5, lo_type=5, pa=1))
    st = Stepper(lib, case, device="cuda:0", fused=False)
    for _ in range(2):
        st.step(case.dt)  # a state with structure in it
    ctx = st.ctx
    ctx.set_bounds_type(args.bt)
    n = case.ne_owned * case.ndof
    ne = case.ne_owned
    dev = st.x.device
    u = st.x
    f64 = dict(dtype=torch.float64, device=dev)
    du_ho, du_lo, du, umin, umax, y = (torch.empty(n, **f64) for _ in range(6))
    xe_min, xe_max = torch.empty(ne, **f64), torch.empty(ne, **f64)
    dt = case.dt
    ctx.setup(st.t)
    ctx.ho_apply(u, du_ho)
    m_ptr = ctx.lumped_mass_ptr()
    vec = 8.0 * n
    sten = 4.0 * 27 * ne
    hob = {1: 8.0 * (2 * 8 + 6 * 4 + 162), 2: 8.0 * (2 * 27 + 6 * 9 + 162)}.get(args.order, 8.0 * (2 * (args.order + 1) ** 3 + 6 * (args.order + 1) ** 2 + 162))
    rows = [
        ("ho_apply (HO alg. bytes)", lambda: ctx.ho_apply(u, du_ho), hob * ne, (du_ho,)),
        ("elem_minmax", lambda: ctx.elem_minmax(u, xe_min, xe_max), vec + 16.0 * ne, (xe_min, xe_max)),
        ("bounds", lambda: ctx.bounds(xe_min, xe_max, umin, umax), 2 * vec + sten + 16.0 * ne, (umin, umax)),
        ("lo_massavg", lambda: ctx.lo_massavg(u, du_ho, dt, du_lo), 4 * vec, (du_lo,)),
        ("fct_clipscale", lambda: ctx.fct_clipscale(u, m_ptr, du_ho, du_lo, umin, umax, dt, du), 7 * vec, (du,)),
        ("limit_fused (du)", lambda: ctx.limit_fused(u, du_ho, dt, du=y), 4 * vec + sten, (y,)),
        ("limit_fused (RK update)", lambda: ctx.limit_fused(u, du_ho, dt, x_base=u, a=0.75, b=0.25, dt_rk=dt, y_out=y),
         5 * vec + sten, (y,)),
    ]
    print(f"== {name}: {args.mesh} -rs {args.rs} -o {args.order}: {ne} elements, {n} dofs, bounds type {args.bt}")
    for label, fn, nbytes, outs in rows:
        ms = timed(fn, args.reps)
        fn()
        torch.cuda.synchronize()
        key = label
        diff = 0.0
        if key in ref:
            diff = max(float((a - b).abs().max()) for a, b in zip(outs, ref[key]))
        else:
            ref[key] = tuple(o.clone() for o in outs)
        print(f"  {label:26s} {1e3 * ms:8.1f} us  {nbytes / (1e9 * ms):7.2f} TB/s  = {nbytes / (1e9 * ms) / 8.0:5.3f} of 8 TB/s"
              f"   max|out - first| {diff:.3e}", flush=True)
    st.close()
    del st, case
    torch.cuda.empty_cache()
