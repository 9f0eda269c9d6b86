"""Development aid: time the one-kernel RK stage with several builds of the library in one process
(tools/build_variant.sh) and compare their final states with the first one.

    python tools/kbench.py [--order 3 --rs 5 --steps 10 --lo 5] name1 name2 ...     (remhos_amd/librmh_<name>.so; "" = librmh.so)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper

ap = argparse.ArgumentParser()
ap.add_argument("--order", type=int, default=3)
ap.add_argument("--rs", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--lo", type=int, default=5)
ap.add_argument("--mesh", default="periodic-cube")
ap.add_argument("--problem", type=int, default=10, help="10: remap (default); 0: transport on a static mesh")
ap.add_argument("--tile", type=int, default=0, help="element numbering of the case builder: y-strips of this many rows (rmhd_config.tile_rows); name suffix ':T' overrides per name")
ap.add_argument("--exact", action="store_true", help="converged local mass solve instead of the -pa rule")
ap.add_argument("names", nargs="+")
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = None
for name in args.names:
    name, _, tl = name.partition(":")
    tile = int(tl) if tl else args.tile
    name, _, form = name.partition("@")  # "main@split": HO kernel + LO kernel + fused limiter instead of the one-kernel stage
    path = os.path.join(root, "remhos_amd", f"librmh_{name}.so" if name not in ("", "main") else "librmh.so")
    lib = bind_driver(load_library(path))
    case = Case(lib, make_config(args.mesh, args.rs, args.order, args.problem, -1.0, 0.5, lo_type=args.lo, pa=0 if args.exact else 1, tile_rows=tile))
    st = Stepper(lib, case, device="cuda:0", one_kernel=(form != "split"))
    for _ in range(2):
        st.step(case.dt)
    st.ctx.enable_timers(True)
    st.ctx.reset_timers()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.step(case.dt)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tim = st.ctx.timers()
    x = st.x.clone()[torch.from_numpy(case.owned_gid.argsort()).to(st.x.device)]  # (compared in global element order)
    it = st.ctx.last_cg_iters()
    mass, umax = st.local_mass_and_max()
    if ref is None:
        ref = x
        diff = 0.0
    else:
        diff = float((x - ref).abs().max())
    nd = case.ne_global * case.ndof
    print(f"{(name or 'main') + ('@' + form if form else '') + (':' + str(tile) if tile else ''):14s} {1e-6 * nd * 3 * args.steps / el:9.1f} MDOFs*stage/s  kernel {1e3 * tim[0] / (3 * args.steps):7.4f} ms (lo {1e3 * tim[2] / (3 * args.steps):.4f} lim {1e3 * tim[3] / (3 * args.steps):.4f})  "
          f"cg {it}  mass {mass:.15g}  max|x - x_first| {diff:.3e}", flush=True)
    st.close()
    del st, case
    torch.cuda.empty_cache()
