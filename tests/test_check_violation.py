"""-vb: check_violation (remhos.cpp:1557-1594, called at :1824-1837 and remhos_fct.cpp:568-610).

rmh_check_violation against its oracle twin on the same inputs, through the C ABI: the verdict on a limited stage (none),
on the UNLIMITED high-order update of the same stage (violations: that is what the limiter is for), with a tolerance, with
the active-dof mask and with the scaled bounds of the product field.  CPU: the kernel sources under the host emulation;
GPU (`-m gpu`): librmh.so."""
import numpy as np
import pytest

from oracle.remhos_oracle import Config, Remhos, check_violation
from tests.helpers import emu_library_path, layout_from_oracle, perturbed


def _case(p, mesh="periodic-cube", prob=0):
    cfg = Config(mesh=mesh, rs=1 if p <= 3 else 0, order=p, problem=prob, dt=0.02, t_final=0.7, lo=5)
    r = Remhos(cfg)
    u = perturbed(r.u)
    keep = {}
    r.stage(u, 0.0, cfg.dt, keep)
    return cfg, r, u, keep


def _same(a, b):
    assert a["count"] == b["count"] and a["first"] == b["first"], (a, b)
    for k in ("over", "under", "first_min", "first_value", "first_max"):
        assert abs(a[k] - b[k]) <= 4e-16 * max(1.0, abs(b[k])), (k, a, b)  # (u + dt du: fused on the device, two roundings in numpy)


def _run(ctx, to_dev, case):
    cfg, r, u, keep = case
    dt = cfg.dt
    d = {k: to_dev(np.ascontiguousarray(keep[k])) for k in ("du", "du_ho", "du_lo", "umin", "umax")}
    ud = to_dev(u)
    # the limited update and the LO update stay inside the bounds (the reference's two calls, remhos.cpp:1824-1837)
    for name in ("du_lo", "du"):
        got = ctx.check_violation(ud, d["umin"], d["umax"], dt=dt, du=d[name])
        assert got["count"] == 0 and got["first"] == -1 and got["over"] == 0.0 and got["under"] == 0.0, (name, got)
        _same(got, check_violation(u, keep["umin"], keep["umax"], dt=dt, du=keep[name]))
    # the unlimited HO update of the same stage does not: the check has to FAIL there
    want = check_violation(u, keep["umin"], keep["umax"], dt=dt, du=keep["du_ho"])
    got = ctx.check_violation(ud, d["umin"], d["umax"], dt=dt, du=d["du_ho"])
    assert want["count"] > 0 and max(want["over"], want["under"]) > 1e-6, want
    _same(got, want)
    # first overload (u_new given), a tolerance that forgives the smaller violations, and a mask
    u_new = u + dt * keep["du_ho"]
    tol = 0.5 * max(want["over"], want["under"])
    w2 = check_violation(u_new, keep["umin"], keep["umax"], tol=tol)
    assert 0 < w2["count"] < want["count"]
    _same(ctx.check_violation(to_dev(u_new), d["umin"], d["umax"], tol=tol), w2)
    mask = (np.arange(u.size).reshape(u.shape) % 3 != want["first"] % 3)
    w3 = check_violation(u_new, keep["umin"], keep["umax"], active_dofs=mask)
    assert w3["first"] != want["first"] and 0 < w3["count"] < want["count"]
    _same(ctx.check_violation(to_dev(u_new), d["umin"], d["umax"], active_dofs=to_dev(mask.astype(np.uint8))), w3)
    # scaled bounds (ScaleProductBounds, remhos_fct.cpp:117-153): (s_min w, s_max w) with a positive weight field w
    w = 1.0 + 0.5 * np.cos(np.arange(u.size, dtype=np.float64).reshape(u.shape))
    w4 = check_violation(u_new * w, keep["umin"], keep["umax"], scale=w, tol=1e-12)
    g4 = ctx.check_violation(to_dev(u_new * w), d["umin"], d["umax"], bound_scale=to_dev(w), tol=1e-12)
    assert w4["count"] > 0
    # (a dof whose violation is within round-off of the tolerance may fall on either side: counts within a few, same first)
    assert abs(g4["count"] - w4["count"]) <= 2 and g4["first"] == w4["first"], (g4, w4)


@pytest.mark.parametrize("p", [2, 3, 4])
def test_check_violation_emu(p):
    from remhos_amd.capi import Context, load_library

    lib = load_library(emu_library_path())
    cfg, r, u, keep = _case(p)
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=r.exec_mode, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    _run(ctx, lambda a: a, (cfg, r, u, keep))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("p", [2, 3, 6])
def test_check_violation_gpu(p):
    import torch

    from remhos_amd.capi import Context, load_library

    lib = load_library()
    cfg, r, u, keep = _case(p)
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=r.exec_mode, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    _run(ctx, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0"), (cfg, r, u, keep))
    ctx.close()


@pytest.mark.gpu
def test_check_violation_on_device_stage():
    """The stage the DEVICE computed: limited update passes, its own unlimited HO rate fails (bounds from the device too)."""
    import torch

    from remhos_amd.capi import Context, load_library

    lib = load_library()
    cfg, r, u_h, keep = _case(3)
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=3, exec_mode=r.exec_mode, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    u = torch.from_numpy(u_h).to("cuda:0")
    du_ho, du, umin, umax = (torch.empty_like(u) for _ in range(4))
    xmn, xmx = (torch.empty(u.shape[0], dtype=torch.float64, device="cuda:0") for _ in range(2))
    ctx.setup(0.0)
    ctx.ho_apply(u, du_ho)
    ctx.limit_fused(u, du_ho, cfg.dt, du=du)
    ctx.elem_minmax(u, xmn, xmx)
    ctx.bounds(xmn, xmx, umin, umax)
    ok = ctx.check_violation(u, umin, umax, dt=cfg.dt, du=du)
    bad = ctx.check_violation(u, umin, umax, dt=cfg.dt, du=du_ho)
    assert ok["count"] == 0, ok
    assert bad["count"] > 0 and bad["first"] >= 0 and max(bad["over"], bad["under"]) > 1e-6, bad
    ctx.close()


@pytest.mark.parametrize("kw", [dict(fused=1), dict(fused=0), dict(fused=1, lo_type=4), dict(fused=1, ps=1, ode_solver=13),
                                dict(fused=0, ps=1, ode_solver=12)],
                         ids=["one-kernel", "sequence", "one-kernel-lo4", "ps-idp3-fused-limiter", "ps-idp2-sequence"])
def test_driver_runs_with_verify_bounds_emu(kw):
    """rmhd_run with -vb (rmhd_config.verify_bounds): the checks of remhos.cpp:1218-1262, 1824-1837 and remhos_fct.cpp:568-610
    run beside every stage (a violation would abort the process) and do not change the result."""
    import ctypes as C

    from remhos_amd.capi import load_library
    from remhos_amd.case import RmhdResult, bind_driver, make_config

    lib = bind_driver(load_library(emu_library_path()))
    out = []
    for vb in (0, 1):
        res = RmhdResult()
        cfg = make_config("cube01_hex", 0, 2, 10, 0.02 if kw.get("ps") else -1.0, 0.5, max_steps=2, verify_bounds=vb, **kw)
        assert lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, lib.rmhd_last_error()
        out.append((res.final_mass, res.max_value, res.final_mass_us, res.steps))
    assert out[0] == out[1]
