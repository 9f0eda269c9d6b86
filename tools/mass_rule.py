"""Development aid: what does a stopping rule of the element-local mass solve cost in accuracy, and what does it buy?

For every rule "rel_tol:abs_tol:max_iter:jacobi_step:constant_mode" (rmh_set_mass_tol / rmh_set_mass_completion; the first
rule is the yardstick, normally the converged solve 1e-14:0:100:0:0):
  * the reference's known answers through the C++ driver (ctest #3, #7: relative deviation of the final mass from the
    reference's 17-digit value; autotest lo 4 runs: do the 10 printed digits of mass / max survive);
  * whole runs at bench size (p = 3 -rs 5, p = 6 -rs 4): throughput, PCG iterations, final mass relative to the yardstick's,
    max-norm distance of the final field to the yardstick's.

    python tools/mass_rule.py [--steps 20] [--small] rule1 rule2 ...
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remhos_amd.capi import load_library
from remhos_amd.case import Case, RmhdResult, bind_driver, make_config
from remhos_amd.stepper import Stepper

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--small", action="store_true", help="skip the bench-size runs")
ap.add_argument("--nokat", action="store_true")
ap.add_argument("--growth", action="store_true", help="drift of two runs that differ by a rounding error")
ap.add_argument("--sizes", default="3:5,6:4", help="order:rs pairs of the bench-size runs")
ap.add_argument("rules", nargs="+")
args = ap.parse_args()
lib = bind_driver(load_library())

KAT = [
    ("ctest3", dict(mesh="cube01_hex", rs=1, order=2, problem=10, dt=-1.0, t_final=0.5, max_steps=5), 0.11972857593296446, None),
    ("ctest7", dict(mesh="cube01_hex", rs=3, order=3, problem=10, dt=-1.0, t_final=0.5, max_steps=1), 0.11601536511552431, None),
    ("auto-tr-lo4", dict(mesh="periodic-cube", rs=1, order=2, problem=0, dt=0.015, t_final=2.0, lo_type=4), 0.9607429525, 0.9334903111),
    ("auto-rm-lo4", dict(mesh="cube01_hex", rs=1, order=2, problem=10, dt=0.02, t_final=0.7, lo_type=4), 0.1197299801, 0.9997499683),
]


def parse(rule):
    rel, ab, it, jac, fix = rule.split(":")
    return float(rel), float(ab), int(it), int(jac), int(fix)


if not args.nokat:
    # (the C++ driver knows two rules: pa = 0 converged, pa = 1 DGMassInverse's rule + completion)
    for pa in (0, 1):
        out = []
        for name, kw, mass, umax in KAT:
            res = RmhdResult()
            cfg = make_config(fused=1, pa=pa, **kw)
            assert lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, lib.rmhd_last_error()
            if umax is None:
                out.append(f"{name} dm {(res.final_mass - mass) / mass:+.2e} it {res.cg_iters_max}")
            else:
                ok = float(f"{res.final_mass:.10g}") == mass and float(f"{res.max_value:.10g}") == umax
                out.append(f"{name} {'10-digit ok' if ok else 'DIGITS LOST'} (max {res.max_value:.12f}) it {res.cg_iters_max}")
        print(f"[kat] pa {pa} " + " | ".join(out), flush=True)

if not args.small:
    for pr in args.sizes.split(","):
        order, rs = (int(v) for v in pr.split(":"))
        case = Case(lib, make_config("periodic-cube", rs, order, 10, -1.0, 0.5))
        ref = None
        for rule in args.rules:
            rel, ab, it, jac, fix = parse(rule)
            st = Stepper(lib, case, device="cuda:0")
            st.ctx.set_mass_tol(rel, ab, it)
            st.ctx.set_mass_completion(jac, fix)
            for _ in range(3):
                st.step(case.dt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                st.step(case.dt)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            mass, umax = st.local_mass_and_max()
            x = st.x.clone()
            cg = st.ctx.last_cg_iters()
            if ref is None:
                ref = (mass, x)
            nd = case.ne_global * case.ndof
            print(f"[run] p {order} rs {rs} {rule:22s} {1e-6 * nd * 3 * args.steps / el:9.1f} MDOFs*stage/s  cg {cg}  mass_rel_dev {(mass - ref[0]) / ref[0]:+.2e}  "
                  f"max|x - x_ref| {float((x - ref[1]).abs().max()):.2e}  max u {umax:.12f}", flush=True)
            st.close()
            del st
        del ref, case
        torch.cuda.empty_cache()

if args.growth:
    # how fast do two runs that differ by a rounding error drift apart?  (yardstick vs yardstick + constant-mode completion)
    for pr in args.sizes.split(","):
        order, rs = (int(v) for v in pr.split(":"))
        case = Case(lib, make_config("periodic-cube", rs, order, 10, -1.0, 0.5))
        a, b = Stepper(lib, case, device="cuda:0"), Stepper(lib, case, device="cuda:0")
        b.ctx.set_mass_completion(0, 1)
        line = []
        for n in range(args.steps + 3):
            a.step(case.dt)
            b.step(case.dt)
            line.append(f"{float((a.x - b.x).abs().max()):.1e}")
        print(f"[growth] p {order} rs {rs} dt {case.dt:.3e} max|x_a - x_b| per step: " + " ".join(line), flush=True)
        a.close(); b.close()
