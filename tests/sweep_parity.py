"""One-off robustness sweep (GPU): the one-kernel stage -- and, with --granular, the HO kernel + LO solver + fused limiter
(rmh_stream.hpp) and the reference's call sequence -- against the oracle over option combinations that the test suite does
not enumerate exhaustively (order x LO solver x bounds type x remap/transport x mesh)."""
import itertools
import sys

import numpy as np
import torch

sys.path.insert(0, ".")  # run from the repository root: python tests/sweep_parity.py
from oracle.remhos_oracle import Config, Remhos  # noqa: E402
from remhos_amd.capi import Context, load_library  # noqa: E402
from tests.helpers import layout_from_oracle, perturbed  # noqa: E402

REL = {1: 1e-12, 2: 1e-12, 3: 5e-10, 4: 5e-9, 5: 1e-7, 6: 2e-7}
lib = load_library()
GRANULAR = "--granular" in sys.argv
COMPLETION = "--completion" in sys.argv  # the converged solve + Jacobi step + constant mode: the same tolerances hold
bad = 0
n = 0
for p, lo, bt, (mesh, prob, rs) in itertools.product((1, 2, 3, 4, 5, 6), (3, 4, 5), (0, 1),
                                                     (("cube01_hex", 10, 1), ("periodic-cube", 0, 0), ("periodic-cube", 10, 0))):
    if lo != 5 and p < 2:
        continue
    if p >= 5 and mesh == "cube01_hex":
        rs = 0
    cfg = Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=0.01, t_final=0.7, lo=lo, bounds_type=bt)
    r = Remhos(cfg)
    r.refine_steps = 2
    x0, vel, nbr, st = layout_from_oracle(r)
    sub = None
    if lo == 4:
        sv = r.Vs if r.exec_mode == 1 else r.vel(r.Xs0)
        sub = np.ascontiguousarray(sv.transpose(0, 2, 1))
    ctx = Context(lib, order=p, exec_mode=r.exec_mode, x0=x0, vel=vel, face_nbr=nbr, stencil27=st, subcell_vel=sub)
    ctx.set_bounds_type(bt)
    if COMPLETION:
        ctx.set_mass_completion(True, True)
    if lo != 5:
        ctx.set_lo_type(lo)
    u_h = perturbed(r.u)
    t = 0.35 if r.exec_mode == 1 else 0.0
    du_ref = r.stage(u_h, t, cfg.dt)
    u = torch.from_numpy(u_h).to("cuda:0")
    y, du = torch.empty_like(u), torch.empty_like(u)
    ctx.setup(t)
    if GRANULAR:
        k, dulo, du2, umin, umax = (torch.empty_like(u) for _ in range(5))
        xmn, xmx = (torch.empty(u.shape[0], dtype=u.dtype, device=u.device) for _ in range(2))
        ctx.ho_apply(u, k)
        if lo == 5:
            ctx.lo_massavg(u, k, cfg.dt, dulo)
            ctx.limit_fused(u, k, cfg.dt, du=du)
        else:
            (ctx.lo_rdsubcell if lo == 4 else ctx.lo_rd)(u, dulo)
            ctx.limit_fused_lo(u, k, dulo, cfg.dt, du=du)
        ctx.elem_minmax(u, xmn, xmx)
        ctx.bounds(xmn, xmx, umin, umax)
        ctx.fct_clipscale(u, ctx.lumped_mass_ptr(), k, dulo, umin, umax, cfg.dt, du2)
        du = torch.where((du - du2).abs() > (du2 - torch.from_numpy(du_ref).to(u.device)).abs(), du, du2)  # the worse of the two
    else:
        ctx.stage_fused(u, cfg.dt, y, du=du, dt_rk=cfg.dt)
    torch.cuda.synchronize()
    err = float(np.abs(du.cpu().numpy() - du_ref).max() / np.abs(du_ref).max())
    n += 1
    flag = "" if err < REL[p] else "   <-- FAIL"
    if flag:
        bad += 1
    print(f"p={p} lo={lo} bt={bt} {mesh:14s} prob={prob:2d} ne={u.shape[0]:3d}  rel err {err:.2e}{flag}")
    ctx.close()
print(f"{n} combinations, {bad} failures")
sys.exit(1 if bad else 0)
