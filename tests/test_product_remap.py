"""Product-field remap (-ps): rmh_product_ratio, rmh_elem_minmax_masked, rmh_fct_product against the
oracle's line-by-line restatement of remhos_sync.cpp / remhos_fct.cpp:26-153, 543-566 on a field with empty zones
(CPU: emulated kernels; GPU twin in tests/test_gpu_product.py), plus the properties the reference states
(remhos_fct.hpp:72-76): bounds preservation of s = us/u and conservation of sum m us."""
import numpy as np
import pytest

from oracle.remhos_oracle import Config, Remhos, compute_ratio, elem_minmax_masked, fct_product
from tests.helpers import layout_from_oracle


def product_case(p, mesh="cube01_hex", rs=1):
    """a density u with empty zones and a product us = s * u with a smooth ratio s in (1, 3)"""
    r = Remhos(Config(mesh=mesh, rs=rs, order=p, problem=10, dt=0.01, t_final=0.7, lo=5))
    ne, nd = r.u.shape
    k = np.arange(ne * nd, dtype=np.float64).reshape(ne, nd)
    u = r.u.copy()
    u[u < 0.2] = 0.0                       # empty dofs and (where the bump is far) empty elements
    u[::5] = 0.0                           # some elements entirely empty
    s_true = 2.0 + np.sin(0.37 * k)
    us = s_true * u
    return r, u, us


def run_product(r, lib, as_dev, to_np, p, u, us, dt, t):
    from remhos_amd.capi import Context

    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    ne, nd = u.shape
    d = {k: as_dev(np.zeros((ne, nd))) for k in ("s", "d_u", "d_us_ho", "smin", "smax", "m", "d_us", "y")}
    flags = {k: as_dev(np.zeros(n, dtype=np.uint8)) for k, n in (("el", ne), ("dofs", ne * nd), ("el2", ne), ("dofs2", ne * nd))}
    xe = {k: as_dev(np.zeros(ne)) for k in ("lo", "hi")}
    ud, usd = as_dev(u), as_dev(us)
    ctx.setup(t)
    # the density's own stage gives u_new = u + dt d_u; the product's HO rate uses the same operator
    ctx.stage_fused(ud, dt, d["y"], du=d["d_u"])
    ctx.ho_apply(usd, d["d_us_ho"])
    ctx.compute_lumped_mass(t, d["m"])
    ctx.product_ratio(usd, ud, d["s"], flags["el"], flags["dofs"])
    ctx.elem_minmax_masked(d["s"], flags["el"], flags["dofs"], xe["lo"], xe["hi"])
    ctx.bounds(xe["lo"], xe["hi"], d["smin"], d["smax"])
    u_new = as_dev(to_np(ud) + dt * to_np(d["d_u"]))
    ctx.product_ratio(None, u_new, None, flags["el2"], flags["dofs2"])
    smin_in, smax_in = to_np(d["smin"]).copy(), to_np(d["smax"]).copy()
    ctx.fct_product(usd, d["m"], d["d_us_ho"], d["smin"], d["smax"], u_new, flags["el2"], flags["dofs2"], dt, d["d_us"])
    out = {k: to_np(v).copy() for k, v in d.items()}
    out.update(el=to_np(flags["el"]).astype(bool), dofs=to_np(flags["dofs"]).astype(bool).reshape(ne, nd),
               el2=to_np(flags["el2"]).astype(bool), dofs2=to_np(flags["dofs2"]).astype(bool).reshape(ne, nd),
               xe_lo=to_np(xe["lo"]).copy(), xe_hi=to_np(xe["hi"]).copy(), u_new=to_np(u_new).copy(),
               smin_in=smin_in, smax_in=smax_in)
    ctx.close()
    return out


def check_product(r, o, u, us, dt, vec_tol):
    # ratio and flags
    s_ref, el_ref, dofs_ref = compute_ratio(us, u)
    assert np.array_equal(o["el"], el_ref) and np.array_equal(o["dofs"], dofs_ref)
    assert np.abs(o["s"] - s_ref).max() < 1e-14
    lo_ref, hi_ref = elem_minmax_masked(o["s"], o["el"], o["dofs"])
    assert np.array_equal(o["xe_lo"], lo_ref) and np.array_equal(o["xe_hi"], hi_ref)
    assert np.isinf(o["xe_lo"][~o["el"]]).all()  # empty elements: the identities of the bounds stencil
    # CalcFCTProduct on the kernel's own inputs
    d_ref, smin_ref, smax_ref = fct_product(us, o["m"], o["d_us_ho"], o["smin_in"], o["smax_in"], o["u_new"], o["el2"], o["dofs2"], dt)
    scale = np.abs(d_ref).max()
    assert np.abs(o["d_us"] - d_ref).max() <= vec_tol * scale
    # (bounds widened to s_avg carry the round-off of the element sums: tree order in the kernel, sequential in the oracle)
    for a, b in ((o["smin"], smin_ref), (o["smax"], smax_ref)):
        assert np.array_equal(np.isfinite(a), np.isfinite(b))
        fin = np.isfinite(b)
        assert np.abs(a[fin] - b[fin]).max() <= 1e-13 and np.array_equal(a[~fin], b[~fin])
    # the properties the reference states (remhos_fct.hpp:72-76), on active dofs of active elements
    us_new = us + dt * o["d_us"]
    on = o["el2"][:, None] & o["dofs2"]
    with np.errstate(invalid="ignore"):  # (inf * 0 on inactive dofs, masked out by `on`)
        lo_b, hi_b = (o["smin"] * o["u_new"])[on], (o["smax"] * o["u_new"])[on]
    assert (us_new[on] + 1e-12 >= lo_b).all() and (us_new[on] - 1e-12 <= hi_b).all()
    act = o["el2"]
    lhs = (o["m"] * (us + dt * o["d_us_ho"])).sum(1)[act]
    rhs = (o["m"] * us_new).sum(1)[act]
    assert np.abs(lhs - rhs).max() <= 1e-13 * np.abs(lhs).max()
    assert (o["d_us"][~act] == 0.0).all()


@pytest.mark.parametrize("p", [2])
def test_product_remap_emulated(p):
    from remhos_amd.capi import load_library
    from tests.helpers import emu_library_path

    lib = load_library(emu_library_path())
    r, u, us = product_case(p)
    o = run_product(r, lib, lambda a: np.ascontiguousarray(a), lambda a: np.asarray(a), p, u, us, 0.01, 0.3)
    assert o["el"].sum() not in (0, len(o["el"]))  # the case has empty and non-empty elements
    check_product(r, o, u, us, 0.01, 1e-12)
