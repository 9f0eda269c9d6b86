"""dim = 2 through the C ABI (rmh_layout.dim = 2, remhos_amd/csrc/rmh_2d.hpp + the dimension-generic streaming kernels):
the HO solver and the granular limiter sequence on quadrilateral lattices, against the oracle stage by stage and against the
REFERENCE's own known answers for whole runs -- the 2-D entries of its ctest table (remhos_tests.cpp:38-107: inline-quad,
-ho 3 -lo 5 -fct 2, full assembly #0-#2 and -pa #4 / #5; #10 is #4 on its CUDA device).

The case (mesh nodes, remap displacement, initial field, CFL step) comes from the oracle's case builder -- test
infrastructure that only prepares INPUTS here; every stage is computed by the library.  CPU: the g++ emulation build of the
same kernel sources; GPU (-m gpu): librmh.so on the device.
"""
import json
import os

import numpy as np
import pytest

from oracle.remhos_oracle import Config, Remhos
from tests.helpers import emu_library_path, layout_from_oracle, perturbed

KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kat.json")))
CTEST = {e["name"].split()[0]: e for e in KAT["ctest"]}


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


class Backend:
    """numpy arrays for the emulation library, torch CUDA tensors for the device library"""

    def __init__(self, gpu):
        self.gpu = gpu
        from remhos_amd.capi import load_library
        from remhos_amd.case import bind_driver

        if gpu:
            import torch

            assert torch.cuda.is_available()
            self.torch = torch
            self.lib = bind_driver(load_library())
        else:
            self.lib = bind_driver(load_library(emu_library_path()))

    def arr(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return self.torch.from_numpy(a).cuda() if self.gpu else a.copy()

    def host(self, a):
        return a.cpu().numpy() if self.gpu else a

    def context(self, r, pa=False):
        from remhos_amd.capi import Context

        x0, vel, nbr, st = layout_from_oracle(r)
        sub = None
        if r.cfg.lo == 4:  # sub-mesh node velocity (v_sub_gf, remhos.cpp:837-853)
            sv = r.Vs if r.exec_mode == 1 else r.vel(r.Xs0)
            sub = np.ascontiguousarray(sv.transpose(0, 2, 1))
        ctx = Context(self.lib, order=r.T.p, exec_mode=r.exec_mode, x0=x0, vel=vel, face_nbr=nbr, stencil27=st, subcell_vel=sub)
        assert ctx.dim == 2
        if self.gpu:
            ctx.set_stream(self.torch.cuda.current_stream().cuda_stream)
        if r.cfg.lo in (3, 4):
            ctx.set_lo_type(r.cfg.lo)
        if pa:
            ctx.set_mass_tol(0.0, 1e-8, 100)  # DGMassInverse's rule (remhos_ho.cpp:79-80) ...
            ctx.set_mass_completion(True, True)  # ... completed (include/remhos_amd/solvers.hpp)
        return ctx


@pytest.fixture(scope="module")
def emu():
    return Backend(False)


@pytest.fixture(scope="module")
def dev():
    return Backend(True)


def check_stage(bk, mesh, rs, p, prob, t, bt=0):
    cfg = Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=0.004, t_final=0.7, lo=5, bounds_type=bt)
    r = Remhos(cfg)
    assert r.dim == 2
    r.refine_steps = 2
    ctx = bk.context(r)
    ctx.set_bounds_type(bt)
    uh = perturbed(r.u)
    keep = {}
    r.stage(uh, t, cfg.dt, keep)
    u = bk.arr(uh)
    z = lambda: bk.arr(np.zeros_like(uh))  # noqa: E731
    du_ho, du, du2, dulo, m, umin, umax, y = (z() for _ in range(8))
    xmn, xmx = bk.arr(np.zeros(r.lat.ne)), bk.arr(np.zeros(r.lat.ne))
    ctx.setup(t)
    ctx.ho_apply(u, du_ho)
    ctx.compute_lumped_mass(t, m)
    ctx.limit_fused(u, du_ho, cfg.dt, du=du)
    ctx.lo_massavg(u, du_ho, cfg.dt, dulo)
    ctx.elem_minmax(u, xmn, xmx)
    ctx.bounds(xmn, xmx, umin, umax)
    ctx.fct_clipscale(u, ctx.lumped_mass_ptr() if bk.gpu else m, du_ho, dulo, umin, umax, cfg.dt, du2)
    ctx.limit_fused(u, du_ho, cfg.dt, x_base=u, a=0.75, b=0.25, dt_rk=cfg.dt, y_out=y)
    h = bk.host
    assert _rel(h(m), keep["m"]) < 1e-13
    tol = {1: 1e-11, 2: 1e-11, 3: 1e-10, 4: 1e-10, 5: 1e-9, 6: 1e-8}[p]  # (conditioning of the Bernstein <-> GL change of basis)
    assert _rel(h(du_ho), keep["du_ho"]) < tol
    assert _rel(h(dulo), keep["du_lo"]) < tol
    assert np.array_equal(h(xmn), uh.min(axis=1)) and np.array_equal(h(xmx), uh.max(axis=1))
    assert np.array_equal(h(umin), keep["umin"]) and np.array_equal(h(umax), keep["umax"])
    assert _rel(h(du), keep["du"]) < tol
    assert _rel(h(du2), keep["du"]) < tol
    assert _rel(h(y), 0.75 * uh + 0.25 * (uh + cfg.dt * keep["du"])) < tol
    assert 0 < ctx.last_cg_iters() < 100
    ctx.close()


def run_case(bk, kw, pa, granular):
    """the time loop of remhos.cpp:1146-1296 (fixed dt) with RK3-SSP, every stage through the C ABI; returns the final
    mass (lumped mass at the final mesh position, remhos.cpp:1394-1413), the field and the oracle object"""
    r = Remhos(Config(**kw))
    ctx = bk.context(r, pa=pa)
    x = bk.arr(r.u)
    z = lambda: bk.arr(np.zeros_like(r.u))  # noqa: E731
    y, k, dulo, umin, umax, du, m = (z() for _ in range(7))
    xmn, xmx = bk.arr(np.zeros(r.lat.ne)), bk.arr(np.zeros(r.lat.ne))
    t_final = 1.0 if r.exec_mode == 1 else kw["t_final"]

    def stage(u, t, dt, x_base, a, b, out):
        ctx.setup(t)
        lo = kw.get("lo", 5)
        if granular == "stage":  # the whole stage behind one entry point (a sequence inside the library for dim = 2)
            if out is u:  # (rmh_stage_fused must not write over its input)
                ctx.stage_fused(u, dt, k, x_base=x_base, a=a, b=b, dt_rk=dt)
                out[...] = k
            else:
                ctx.stage_fused(u, dt, out, x_base=x_base, a=a, b=b, dt_rk=dt)
            return
        ctx.ho_apply(u, k)
        if not granular:
            if lo == 5:
                ctx.limit_fused(u, k, dt, x_base=x_base, a=a, b=b, dt_rk=dt, y_out=out)
            else:
                (ctx.lo_rdsubcell if lo == 4 else ctx.lo_rd)(u, dulo)
                ctx.limit_fused_lo(u, k, dulo, dt, x_base=x_base, a=a, b=b, dt_rk=dt, y_out=out)
            return
        if lo == 5:
            ctx.lo_massavg(u, k, dt, dulo)
        else:
            (ctx.lo_rdsubcell if lo == 4 else ctx.lo_rd)(u, dulo)
        ctx.elem_minmax(u, xmn, xmx)
        ctx.bounds(xmn, xmx, umin, umax)
        if bk.gpu:
            ctx.fct_clipscale(u, ctx.lumped_mass_ptr(), k, dulo, umin, umax, dt, du)
        else:
            ctx.compute_lumped_mass(t, m)
            ctx.fct_clipscale(u, m, k, dulo, umin, umax, dt, du)
        ynew = u + dt * du
        out[...] = b * ynew if x_base is None else a * x_base + b * ynew

    t, steps, dt0 = 0.0, 0, r.dt
    while True:
        dt = min(dt0, t_final - t)
        stage(x, t, dt, None, 0.0, 1.0, y)
        stage(y, t + dt, dt, x, 0.75, 0.25, y)
        stage(y, t + dt / 2, dt, x, 1.0 / 3.0, 2.0 / 3.0, x)
        t += dt
        steps += 1
        if t >= t_final - 1e-8 * dt0 or steps == kw.get("max_steps", -1):
            break
    ctx.compute_lumped_mass(t, m)
    xh, mh = bk.host(x), bk.host(m)
    ctx.close()
    return float((mh * xh).sum()), xh, steps, r


def check_rd(bk, mesh, rs, p, prob, t, lo):
    """PAResidualDistribution[Subcell]::CalcLOSolution (remhos_lo.cpp:965-1034, 1620-1802) in 2-D against the oracle's
    restatement of the host form (remhos_lo.cpp:111-245); no linear solve involved"""
    cfg = Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=0.004, t_final=0.7, lo=lo)
    r = Remhos(cfg)
    ctx = bk.context(r)
    uh = perturbed(r.u)
    keep = {}
    r.stage(uh, t, cfg.dt, keep)
    u, dulo = bk.arr(uh), bk.arr(np.zeros_like(uh))
    ctx.setup(t)
    (ctx.lo_rdsubcell if lo == 4 else ctx.lo_rd)(u, dulo)
    assert _rel(bk.host(dulo), keep["du_lo"]) < 1e-12
    ctx.close()


def _kw(e):
    return {k: e[k] for k in ("mesh", "rs", "order", "problem", "dt", "t_final", "lo", "max_steps") if k in e}


# ---- CPU: the emulation build ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mesh,rs,p,prob,t,bt", [("inline-quad", 1, 2, 14, 0.3, 0), ("inline-quad", 0, 3, 14, 0.6, 1),
                                                ("periodic-square", 1, 1, 5, 0.0, 0), ("periodic-square", 0, 4, 5, 0.0, 0),
                                                ("inline-quad", 0, 6, 14, 0.2, 0)])
def test_2d_stage_vs_oracle_emulated(emu, mesh, rs, p, prob, t, bt):
    check_stage(emu, mesh, rs, p, prob, t, bt)


@pytest.mark.parametrize("mesh,rs,p,prob,t,lo", [("inline-quad", 0, 2, 14, 0.3, 4), ("periodic-square", 1, 3, 5, 0.0, 4),
                                                ("inline-quad", 0, 4, 14, 0.6, 4), ("inline-quad", 0, 3, 14, 0.5, 3)])
def test_2d_rd_vs_oracle_emulated(emu, mesh, rs, p, prob, t, lo):
    check_rd(emu, mesh, rs, p, prob, t, lo)


@pytest.mark.parametrize("fused,pa,lo", [(1, 1, 5), (0, 0, 4)])
def test_2d_cpp_driver_emulated(emu, fused, pa, lo):
    """rmhd_run (the C++ restatement of remhos(): case builder build_case_2d, solver classes, RK3 loop, report) on the first
    step of ctest #0's case (= #4 / #9 / #10 with -pa) through the emulated kernels: one-call stages (rmh_stage_fused) with the
    -pa mass rule / the solver classes' call sequence with lo 4 -- mass and maximum against the oracle, which reproduces the
    reference's constants for the whole runs (tests/test_oracle_kat.py).  The five-step runs against the constants themselves
    are the GPU tests below (100 s per run under the OS-thread emulation)."""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    kw = dict(_kw(CTEST["ctest0"]), max_steps=1, lo=lo)
    cfg = make_config(kw["mesh"], kw["rs"], kw["order"], kw["problem"], kw["dt"], kw["t_final"], max_steps=1, lo_type=lo, fused=fused, pa=pa)
    res = RmhdResult()
    assert emu.lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, emu.lib.rmhd_last_error()
    out = Remhos(Config(**kw)).run()
    assert res.steps == out["steps"] == 1
    assert abs(res.final_mass - out["mass"]) <= 1e-13 * abs(out["mass"]) and abs(res.max_value - out["max"]) <= 1e-12


def test_2d_refusals(emu):
    """what dim = 2 does not have says so (no silent 3-D kernel on 2-D data)"""
    r = Remhos(Config(mesh="inline-quad", rs=0, order=2, problem=14, dt=0.01, t_final=0.5, lo=5))
    ctx = emu.context(r)
    u = emu.arr(r.u)
    from remhos_amd.capi import RmhError

    for call in (lambda: ctx.stage_fused_range(u, 0.01, u * 0.0, 0, 1, True), lambda: ctx.halo_pack_records(u, None, 0, u * 0.0)):
        with pytest.raises(RmhError, match="dim = 2"):
            call()
    ctx.close()


# ---- GPU -------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("mesh,rs,p,prob,t,bt", [("inline-quad", 2, 1, 14, 0.4, 0), ("inline-quad", 2, 2, 14, 0.3, 1),
                                                ("inline-quad", 2, 3, 14, 0.6, 0), ("periodic-square", 2, 3, 5, 0.0, 0),
                                                ("inline-quad", 1, 4, 14, 0.5, 0), ("periodic-square", 1, 5, 5, 0.0, 1),
                                                ("inline-quad", 1, 6, 14, 0.2, 0)])
def test_2d_stage_vs_oracle_gpu(dev, mesh, rs, p, prob, t, bt):
    check_stage(dev, mesh, rs, p, prob, t, bt)


@pytest.mark.gpu
@pytest.mark.parametrize("name,pa,granular", [("ctest0", False, True), ("ctest0", True, False), ("ctest1", False, False),
                                              ("ctest2", False, False), ("ctest2", True, True), ("ctest5", True, False),
                                              ("ctest5", True, True), ("ctest0", True, "stage"), ("ctest1", False, "stage")])
def test_2d_reference_ctests_gpu(dev, name, pa, granular):
    """The 2-D entries of the reference's test table on the MI355X: final mass against the reference's 17-digit constants
    (its own check: 10 eps relative to 1 + |x|, remhos_tests.cpp:13-23; BASELINE.json: 1e-12 relative).  pa = False: the
    element-local solve converged (the exact inverse of the full-assembly entries #0-#2); pa = True: DGMassInverse's rule +
    completion (the -pa entries #4 / #5; #2's mesh as a further -pa case)."""
    e = CTEST[name]
    mass, x, steps, r = run_case(dev, _kw(e), pa, granular)
    assert steps == e["max_steps"]
    assert abs(mass - e["mass"]) <= 1e-12 * abs(e["mass"]), (mass, e["mass"])
    assert abs(mass - e["mass"]) <= 5e-14 * (1.0 + abs(e["mass"]))


@pytest.mark.gpu
@pytest.mark.parametrize("mesh,rs,p,prob,t,lo", [("inline-quad", 1, 2, 14, 0.3, 4), ("periodic-square", 2, 3, 5, 0.0, 4),
                                                ("inline-quad", 1, 3, 14, 0.6, 4), ("inline-quad", 1, 4, 14, 0.2, 4),
                                                ("periodic-square", 1, 6, 5, 0.0, 4), ("inline-quad", 1, 3, 14, 0.5, 3)])
def test_2d_rd_vs_oracle_gpu(dev, mesh, rs, p, prob, t, lo):
    check_rd(dev, mesh, rs, p, prob, t, lo)


AUTOTEST_2D = [e for e in KAT["autotest"] if e["mesh"] in ("inline-quad", "periodic-square") and e.get("fct", 2) == 2 and e["ho"] == 3]


@pytest.mark.gpu
@pytest.mark.parametrize("e", AUTOTEST_2D, ids=[e["name"] for e in AUTOTEST_2D])
@pytest.mark.parametrize("granular", [False, True, "stage"])
def test_2d_reference_autotest_gpu(dev, e, granular):
    """The 2-D lines of the reference's regression baseline for -ho 3 -lo 4 -fct 2 (autotest/out_baseline.dat:41-44 inline-quad
    remap, 500 steps; :61-64 periodic-square transport, 200 steps -- BASELINE.json configs[0]'s case): final mass and maximum,
    10 significant digits as the reference prints and diffs them, on the MI355X."""
    kw = {k: e[k] for k in ("mesh", "rs", "order", "problem", "dt", "t_final", "lo")}
    mass, x, steps, r = run_case(dev, kw, False, granular)
    assert float(f"{mass:.10g}") == e["mass"], (mass, e["mass"])
    assert float(f"{x.max():.10g}") == e["max"], (x.max(), e["max"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,pa", [("ctest0", 0), ("ctest0", 1), ("ctest1", 0), ("ctest2", 0), ("ctest5", 1)])
@pytest.mark.parametrize("fused", [1, 0])
def test_2d_cpp_driver_reference_ctests_gpu(dev, name, pa, fused):
    """The same constants through the product's own harness -- rmhd_run: C++ case builder (build_case_2d: the mesh, the remap
    displacement, the initial field, the CFL step), solver classes, RK3 loop, report -- no oracle anywhere in the run.  What
    `remhos_amd_run -m data/inline-quad.mesh -p 14 -rs 4 -o 3 -dt -1 -tf 0.5 -ms 5 -ho 3 -lo 5 -fct 2 [-pa]` prints."""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    e = CTEST[name]
    cfg = make_config(e["mesh"], e["rs"], e["order"], e["problem"], e["dt"], e["t_final"], max_steps=e["max_steps"], fused=fused, pa=pa)
    res = RmhdResult()
    assert dev.lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, dev.lib.rmhd_last_error()
    assert res.steps == e["max_steps"]
    assert abs(res.final_mass - e["mass"]) <= 5e-14 * (1.0 + abs(e["mass"])), (res.final_mass, e["mass"])


@pytest.mark.gpu
@pytest.mark.parametrize("e", AUTOTEST_2D, ids=[e["name"] for e in AUTOTEST_2D])
@pytest.mark.parametrize("fused", [1, 0])
def test_2d_cpp_driver_reference_autotest_gpu(dev, e, fused):
    """autotest/out_baseline.dat:41-44, 61-64 (-ho 3 -lo 4 -fct 2 in 2-D; periodic-square is BASELINE configs[0]'s case) through
    rmhd_run: mass and maximum, ten printed digits"""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    cfg = make_config(e["mesh"], e["rs"], e["order"], e["problem"], e["dt"], e["t_final"], lo_type=4, fused=fused)
    res = RmhdResult()
    assert dev.lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, dev.lib.rmhd_last_error()
    assert float(f"{res.final_mass:.10g}") == e["mass"] and float(f"{res.max_value:.10g}") == e["max"], (res.final_mass, res.max_value)


AUTOTEST_2D_HO2 = [e for e in KAT["autotest"] if e["mesh"] in ("inline-quad", "periodic-square") and e["ho"] == 2]


@pytest.mark.gpu
@pytest.mark.parametrize("e", AUTOTEST_2D_HO2, ids=[e["name"] for e in AUTOTEST_2D_HO2])
@pytest.mark.parametrize("fused", [1, 0])
def test_2d_cpp_driver_reference_autotest_ho2_lo3_gpu(dev, e, fused):
    """autotest/out_baseline.dat:78-81, 98-101: -ho 2 -lo 3 -fct 2 -pa in 2-D (CGHOSolver = the local solve to rel 1e-12,
    plain PAResidualDistribution) through rmhd_run on the MI355X: mass and maximum, ten printed digits"""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    cfg = make_config(e["mesh"], e["rs"], e["order"], e["problem"], e["dt"], e["t_final"], lo_type=e["lo"], ho_type=2, pa=1, fused=fused)
    res = RmhdResult()
    assert dev.lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, dev.lib.rmhd_last_error()
    assert float(f"{res.final_mass:.10g}") == e["mass"] and float(f"{res.max_value:.10g}") == e["max"], (res.final_mass, res.max_value)


@pytest.mark.gpu
@pytest.mark.parametrize("mesh,prob,lo,dt,tf,fused", [("inline-quad", 14, 4, 0.06, 0.75, 1), ("inline-quad", 14, 4, 0.06, 0.75, 0),
                                                      ("periodic-square", 5, 4, 0.02, 0.3, 1), ("periodic-square", 5, 5, 0.02, 0.3, 0)])
def test_2d_cpp_driver_bounds_type_1_and_dt_control_gpu(dev, mesh, prob, lo, dt, tf, fused):
    """-bt 1 -dtc 1 (remhos.cpp:1178-1197, 1968-1998; remhos_tools.cpp:381-430) in 2-D through rmhd_run against the oracle:
    the same accepted and repeated steps, the same final dt, mass and maximum.  (The reference's own auto-dt baselines run
    -fct 4, which is restated in the oracle only; tests/test_oracle_kat.py pins these options there.)"""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    r = Remhos(Config(mesh=mesh, rs=1, order=3, problem=prob, dt=dt, t_final=tf, lo=lo, fct=2, bounds_type=1, dt_control=1))
    out = r.run()
    cfg = make_config(mesh, 1, 3, prob, dt, tf, lo_type=lo, fused=fused, bounds_type=1, dt_control=1)
    res = RmhdResult()
    assert dev.lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, dev.lib.rmhd_last_error()
    assert (res.steps, res.repeats) == (out["steps"], r.repeats), (res.steps, res.repeats, out["steps"], r.repeats)
    assert abs(res.dt - out["dt"]) <= 1e-12 * out["dt"]
    assert abs(res.final_mass - out["mass"]) <= 1e-12 * abs(out["mass"]) and abs(res.max_value - out["max"]) <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [1, 0])
def test_2d_configs0_order_2_as_worded_gpu(dev, fused):
    """BASELINE.json configs[0] exactly as worded -- "2D periodic-square transport, p=2, RK3": `-m data/periodic-square.mesh -p 5
    -rs 3 -o 2 -dt 0.004 -tf 0.8 -ho 3 -lo 4 -fct 2`, 200 steps -- through rmhd_run on the MI355X: the ten digits of SURVEY.md
    Appendix E's cross-check table (tests/golden/reference_kat.json "survey_cross_checks"; the oracle reproduces them in
    tests/test_oracle_kat.py).  The autotest line of the same mesh (-o 3, out_baseline.dat:61-64) is
    test_2d_cpp_driver_reference_autotest_gpu."""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    e = next(x for x in KAT["survey_cross_checks"] if x["mesh"] == "periodic-square" and x["order"] == 2)
    cfg = make_config(e["mesh"], e["rs"], e["order"], e["problem"], e["dt"], e["t_final"], lo_type=e["lo"], fused=fused)
    res = RmhdResult()
    assert dev.lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, dev.lib.rmhd_last_error()
    assert res.steps == 200
    assert float(f"{res.final_mass:.10g}") == e["mass"] and float(f"{res.max_value:.10g}") == e["max"], (res.final_mass, res.max_value)
