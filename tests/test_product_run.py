"""Product-field remap (-ps) end to end through the C++ driver (rmhd_run_state: AdvectionOperator with the block vector
[u | us], IDP solvers of remhos_solvers.cpp) against the oracle, whose product functions are pinned by the reference's
"Product remap 2D IDP3" known answer (tests/test_oracle_kat.py::test_product_remap_idp3).  CPU: the kernel sources under
the host emulation on a tiny 3-D mesh; the GPU twin is tests/test_gpu_product_run.py."""
import ctypes as C

import numpy as np
import pytest


def run_both(lib, mesh, rs, p, dt, ode, steps, fused, tol_field):
    from oracle.remhos_oracle import Config, Remhos
    from remhos_amd.case import RmhdResult, make_config

    r = Remhos(Config(mesh=mesh, rs=rs, order=p, problem=10, dt=dt, t_final=0.5, lo=5, fct=2, ps=True, ode=ode, max_steps=steps))
    r.refine_steps = 2 if p >= 4 else 0
    out = r.run()
    cfg = make_config(mesh, rs, p, 10, dt, 0.5, max_steps=steps, fused=fused, ps=1, ode_solver=ode)
    res = RmhdResult()
    n = r.u.size
    u, us = np.zeros(n), np.zeros(n)
    rc = lib.rmhd_run_state(C.byref(cfg), C.byref(res), u.ctypes.data, us.ctypes.data)
    assert rc == 0, lib.rmhd_last_error()
    assert res.steps == out["steps"] and res.stages == {11: 1, 12: 2, 13: 3}[ode] * steps
    assert abs(res.mass0 - out["mass0"]) <= 1e-14 * abs(out["mass0"])
    assert abs(res.mass0_us - r.mass0_us) <= 1e-14 * abs(r.mass0_us)
    assert abs(res.final_mass - out["mass"]) <= 1e-12 * abs(out["mass"])
    assert abs(res.final_mass_us - out["mass_us"]) <= 1e-12 * abs(out["mass_us"])
    du, dus = np.abs(u.reshape(r.u.shape) - r.u).max(), np.abs(us.reshape(r.u.shape) - r.us).max()
    print(mesh, rs, p, ode, "fused" if fused else "sequence", "field dev", du, dus, "s_max", res.s_max, out["s_max"])
    # us: the ratio s = us / u of a barely active dof (u just above EMPTY_ZONE_TOL = 1e-12) magnifies the rounding
    # differences of us there by 1 / u; through the bounds of s they reach dofs with u = O(1) -- two orders of slack
    assert du < tol_field and dus < 100 * tol_field
    assert abs(res.s_max - out["s_max"]) < 1e3 * tol_field
    # the state has empty zones and active ones (otherwise the masks are not exercised)
    assert (r.u <= 1e-12).any() and (r.u > 1e-12).any()
    return res


@pytest.fixture(scope="module")
def emulib():
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver
    from tests.helpers import emu_library_path

    return bind_driver(load_library(emu_library_path()))


@pytest.mark.parametrize("ode,fused,steps", [(13, 1, 1), (12, 0, 1), (11, 1, 2)])
def test_product_remap_emulated_vs_oracle(emulib, ode, fused, steps):
    run_both(emulib, "cube01_hex", 0, 2, 0.02, ode, steps + 1, fused, 1e-11)
