"""GPU twin of tests/test_product_run.py: -ps runs through rmhd_run_state on the MI355X (HO kernel on u and us, fused or
granular limiter for u, product_ratio / elem_minmax_masked / bounds / fct_product kernels for us, IDP RK solvers) against
the pinned oracle: masses of u and us to 1e-12 relative, the fields themselves, the maximum of s."""
import pytest

from tests.test_product_run import run_both

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import torch

    assert torch.cuda.is_available()
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver

    return bind_driver(load_library())


@pytest.mark.parametrize("mesh,rs,p,dt,ode,steps,fused,tol", [
    ("cube01_hex", 2, 2, 0.02, 13, 6, 1, 1e-11),
    ("cube01_hex", 2, 3, 0.02, 13, 4, 0, 1e-10),
    ("cube01_hex", 1, 3, 0.02, 12, 5, 1, 1e-10),
    ("cube01_hex", 1, 2, 0.02, 11, 8, 0, 1e-11),
    ("periodic-cube", 1, 3, 0.02, 13, 3, 1, 1e-10),
    ("cube01_hex", 1, 4, 0.02, 13, 2, 1, 1e-9),
])
def test_product_remap_vs_oracle(lib, mesh, rs, p, dt, ode, steps, fused, tol):
    res = run_both(lib, mesh, rs, p, dt, ode, steps, fused, tol)
    assert res.fom_wall > 0


def test_product_remap_conserves_and_bounds_s(lib):
    """the two properties the reference states for CalcFCTProduct (remhos_fct.hpp:72-76) over a whole run at a size the
    oracle does not reach: the mass of us changes only by the remap's own time-discretisation drift (like u's), and
    s = us / u stays within the initial range of s0 (2 +- 1) on the active dofs"""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    cfg = make_config("cube01_hex", 3, 3, 10, 0.02, 0.5, max_steps=10, ps=1, ode_solver=13, pa=1)
    res = RmhdResult()
    assert lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, lib.rmhd_last_error()
    assert res.steps == 10 and res.stages == 30
    assert res.mass_loss_us < 1e-6 * res.mass0_us and res.mass_loss < 1e-6 * res.mass0
    assert 1.0 - 1e-9 <= res.s_max <= 3.0 + 1e-9
