"""The HIP kernel sources compiled with g++ against tests/emu (host emulation of workgroups)
and checked against the oracle on tiny meshes: catches indexing/synchronisation bugs and lets
the kernels run under host sanitizers.  The product never loads this library."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.remhos_oracle import Config, Remhos
from tests.helpers import emu_library_path, layout_from_oracle, perturbed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver

    return bind_driver(load_library(emu_library_path()))


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("mesh,rs,p,prob,t", [("cube01_hex", 0, 2, 10, 0.3), ("periodic-cube", 0, 3, 0, 0.0),
                                             ("periodic-cube", 0, 1, 10, 0.6)])
def test_kernels_vs_oracle(lib, mesh, rs, p, prob, t):
    from remhos_amd.capi import Context

    cfg = Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=0.02, t_final=0.7, lo=5)
    r = Remhos(cfg)
    r.refine_steps = 2
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=r.exec_mode, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    u = perturbed(r.u)
    keep = {}
    r.stage(u, t, cfg.dt, keep)
    z = lambda: np.zeros_like(u)
    du_ho, du, du2, dulo, m, umin, umax = (z() for _ in range(7))
    xmn, xmx = np.zeros(r.lat.ne), np.zeros(r.lat.ne)
    ctx.setup(t)
    ctx.ho_apply(u, du_ho)
    ctx.compute_lumped_mass(t, m)
    ctx.limit_fused(u, du_ho, cfg.dt, du=du)
    ctx.lo_massavg(u, du_ho, cfg.dt, dulo)
    ctx.elem_minmax(u, xmn, xmx)
    ctx.bounds(xmn, xmx, umin, umax)
    ctx.fct_clipscale(u, m, du_ho, dulo, umin, umax, cfg.dt, du2)
    assert _rel(m, keep["m"]) < 1e-13
    assert _rel(du_ho, keep["du_ho"]) < 1e-10
    assert _rel(dulo, keep["du_lo"]) < 1e-10
    assert np.array_equal(umin, keep["umin"]) and np.array_equal(umax, keep["umax"])
    assert _rel(du, keep["du"]) < 1e-10
    assert _rel(du2, keep["du"]) < 1e-10
    ctx.close()


@pytest.mark.parametrize("p,bt", [(1, 1), (2, 0), (3, 1), (4, 0), (4, 1), (5, 0), (6, 0)])
def test_streaming_kernels_every_order(lib, p, bt):
    """The wavefront-per-element streaming kernels (rmh_stream.hpp: element extrema, bounds, fused limiter) at every order --
    one to six dof rounds per wavefront, 1 ... 8 elements per pass, a last pass that is not full (27 elements), both bounds
    types, the limiter with its own mass-based average and with a given LO rate -- against the oracle's bounds (bit for bit)
    and against the granular ClipScale kernel on the same inputs.  No HO kernel: du_HO is synthetic."""
    from remhos_amd.capi import Context

    cfg = Config(mesh="periodic-cube", rs=0, order=p, problem=0, dt=0.02, t_final=0.7, lo=5, bounds_type=bt)
    r = Remhos(cfg)
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=r.exec_mode, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    ctx.set_bounds_type(bt)
    rng = np.random.default_rng(7 + p)
    u = perturbed(r.u)
    ne, nd = u.shape
    xmn, xmx = np.zeros(ne), np.zeros(ne)
    ctx.elem_minmax(u, xmn, xmx)
    assert np.array_equal(xmn, u.min(axis=1)) and np.array_equal(xmx, u.max(axis=1))
    umin, umax = np.zeros_like(u), np.zeros_like(u)
    ctx.bounds(xmn, xmx, umin, umax)
    omin, omax = r.bounds_from_extrema(xmn, xmx)
    assert np.array_equal(umin, omin.reshape(u.shape)) and np.array_equal(umax, omax.reshape(u.shape))
    # limiter: HO rate = anything; the lumped mass and the state flag come from rmh_ho_apply
    ctx.setup(0.0)
    du_ho = np.zeros_like(u)
    ctx.ho_apply(u, du_ho)
    du_ho += 0.5 * rng.standard_normal(u.shape)
    m = np.zeros_like(u)
    ctx.compute_lumped_mass(0.0, m)
    dulo, du_ref, du, y = (np.zeros_like(u) for _ in range(4))
    ctx.lo_massavg(u, du_ho, cfg.dt, dulo)
    ctx.fct_clipscale(u, m, du_ho, dulo, umin, umax, cfg.dt, du_ref)
    ctx.limit_fused(u, du_ho, cfg.dt, du=du)
    assert _rel(du, du_ref) < 1e-12
    xb = rng.standard_normal(u.shape)
    ctx.limit_fused(u, du_ho, cfg.dt, x_base=xb, a=0.75, b=0.25, dt_rk=0.5 * cfg.dt, y_out=y)
    assert _rel(y, 0.75 * xb + 0.25 * (u + 0.5 * cfg.dt * du_ref)) < 1e-12
    # a given LO rate (lo 3 / 4 path of the fused limiter)
    dulo2 = dulo * (1.0 + 0.1 * rng.random(u.shape))
    ctx.fct_clipscale(u, m, du_ho, dulo2, umin, umax, cfg.dt, du_ref)
    ctx.limit_fused_lo(u, du_ho, dulo2, cfg.dt, du=du)
    assert _rel(du, du_ref) < 1e-12
    ctx.close()


def test_split_columns_p6_emulated(lib):
    """p = 6: two wavefronts per element; the second one runs the face rows and the 17 quadrature columns beyond the first 64
    with three lanes per column (a third of the qz range each, DPP row-shift sums, w detJ of those columns in LDS).  HO kernel
    and one-kernel stage against the oracle under the host emulation (GPU twins: tests/test_gpu_parity.py at p = 6)."""
    from remhos_amd.capi import Context

    p, t = 6, 0.3
    cfg = Config(mesh="cube01_hex", rs=0, order=p, problem=10, dt=0.01, t_final=0.7, lo=5)
    r = Remhos(cfg)
    r.refine_steps = 2
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    u = perturbed(r.u)
    keep = {}
    du_ref = r.stage(u, t, cfg.dt, keep)
    du_ho, y, du = np.zeros_like(u), np.zeros_like(u), np.zeros_like(u)
    ctx.setup(t)
    ctx.ho_apply(u, du_ho)
    tol = 1e-7  # (p = 6, the same as tests/test_gpu_parity.py: the local mass matrices are badly conditioned)
    assert _rel(du_ho, keep["du_ho"]) < tol
    ctx.stage_fused(u, cfg.dt, y, dt_rk=cfg.dt, du=du)
    assert _rel(du, du_ref) < tol
    assert _rel(y, u + cfg.dt * du_ref) < tol
    ctx.close()


def test_rd_one_element_workgroups_emulated(lib):
    """Subcell residual distribution (lo 4) where a workgroup holds ONE element (p = 4: one wavefront): element extrema and
    the sums of the subcell fluctuations come from thread / DPP reductions, the dofs gather their subcells without branches.
    RD-only kernel and one-kernel lo 4 stage against the oracle under the host emulation (GPU twins:
    tests/test_gpu_parity.py::test_lo_rdsubcell_parity, ::test_one_kernel_stage_lo4)."""
    from remhos_amd.capi import Context

    p, t = 4, 0.3
    cfg = Config(mesh="cube01_hex", rs=0, order=p, problem=10, dt=0.01, t_final=0.7, lo=4)
    r = Remhos(cfg)
    r.refine_steps = 2
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st,
                  subcell_vel=np.ascontiguousarray(r.Vs.transpose(0, 2, 1)))
    u = perturbed(r.u)
    keep = {}
    du_ref = r.stage(u, t, cfg.dt, keep)
    du_lo, y, du = np.zeros_like(u), np.zeros_like(u), np.zeros_like(u)
    ctx.setup(t)
    ctx.lo_rdsubcell(u, du_lo)
    assert _rel(du_lo, keep["du_lo"]) < 1e-12
    ctx.set_lo_type(4)
    ctx.stage_fused(u, cfg.dt, y, dt_rk=cfg.dt, du=du)
    assert _rel(du, du_ref) < 5e-10  # (tests/test_gpu_parity.py REL[4])
    ctx.close()


def test_cpp_driver_and_stepper_vs_oracle(lib):
    """remhos() restated in C++ (rmhd_run: solver classes + RK3) and the Python stepper give the
    oracle's final mass / max after two RK3 steps of a remap."""
    from remhos_amd.case import Case, RmhdResult, make_config
    from remhos_amd.stepper import Stepper

    mesh, rs, p, prob, dt, tf = "cube01_hex", 0, 2, 10, -1.0, 0.5
    r = Remhos(Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=dt, t_final=tf, lo=5, max_steps=2))
    out = r.run()
    for fused in (1, 0):
        res = RmhdResult()
        cfg = make_config(mesh, rs, p, prob, dt, tf, max_steps=2, fused=fused)
        assert lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0
        assert res.steps == 2 and res.stages == 6
        assert abs(res.final_mass - out["mass"]) < 1e-14
        assert abs(res.max_value - out["max"]) < 1e-13
        assert abs(res.mass0 - out["mass0"]) < 1e-15
        st = Stepper(lib, Case(lib, cfg), device="cpu", fused=bool(fused))
        st.run(max_steps=2)
        mass, umax = st.local_mass_and_max()
        assert abs(mass - out["mass"]) < 1e-14
        assert np.abs(st.x.numpy() - r.u).max() < 1e-13
        st.close()


@pytest.mark.parametrize("p", [5])
def test_generic_orders_multi_block_race_free(lib, p):
    """Orders whose dof count is not a multiple of the wavefront size take the generic (chunk sum)
    reductions and different LDS overlays (p = 4: two elements per 128-thread workgroup, p = 5: four per
    256-thread workgroup); several workgroups per launch, repeated launches must agree bit
    for bit (the OS-thread emulation perturbs the interleaving, which exposed an LDS overlap race once)."""
    from remhos_amd.capi import Context

    cfg = Config(mesh="cube01_hex", rs=0, order=p, problem=10, dt=0.01, t_final=0.7, lo=5)
    r = Remhos(cfg)
    r.refine_steps = 2
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    u = perturbed(r.u)
    keep = {}
    r.stage(u, 0.3, cfg.dt, keep)
    ctx.setup(0.3)
    outs = []
    dh = np.zeros_like(u)
    ctx.ho_apply(u, dh)
    for _ in range(2):
        y, du = np.zeros_like(u), np.zeros_like(u)
        ctx.stage_fused(u, cfg.dt, y, du=du)
        outs.append((dh, du))
    tol = {4: 1e-10, 5: 1e-8}[p]
    assert _rel(outs[0][0], keep["du_ho"]) < tol
    assert _rel(outs[0][1], keep["du"]) < tol
    for dh, du in outs[1:]:
        assert np.array_equal(dh, outs[0][0]) and np.array_equal(du, outs[0][1])
    ctx.close()


def test_two_blocks_manual_exchange_both_ghost_layouts(lib):
    """Two blocks of one mesh in one process, ghosts filled by hand: the separate-array ghost interface
    (rmh_halo_pack + rmh_set_ghost_u + rmh_set_ghost_minmax, the shape of MFEM's FaceNbrData) and the
    record interface (rmh_halo_pack_records + rmh_set_ghost_records) give the same stage, which equals
    the single-block stage bit for bit -- whole, and as interior / halo element ranges."""
    from remhos_amd.capi import Context
    from remhos_amd.case import Case, make_config

    mesh, rs, p, prob, t, dt = "cube01_hex", 1, 1, 10, 0.3, 0.01  # 4^3 elements, two blocks of 2 x 4 x 4
    g = Case(lib, make_config(mesh, rs, p, prob, -1.0, 0.5))
    u_g = perturbed(g.u0)

    def mk(c):
        return Context(lib, order=p, exec_mode=c.exec_mode, x0=c.x0, vel=c.vel, face_nbr=c.face_nbr,
                       stencil27=c.stencil27, ne_ghost=c.ne_ghost)

    cg = mk(g)
    cg.setup(t)
    y_g = np.zeros_like(u_g)
    cg.stage_fused(u_g, dt, y_g)
    cases = [Case(lib, make_config(mesh, rs, p, prob, -1.0, 0.5, part=(2, 1, 1), rank=k)) for k in range(2)]
    nd = g.ndof
    for layout in ("arrays", "records"):
        us = [np.ascontiguousarray(u_g[c.owned_gid]) for c in cases]
        ctxs = [mk(c) for c in cases]
        ghosts = []
        for c, ctx in zip(cases, ctxs):
            if layout == "arrays":
                gh = (np.zeros((c.ne_ghost, nd)), np.zeros(c.ne_ghost), np.zeros(c.ne_ghost))
                ctx.set_ghost_u(gh[0])
                ctx.set_ghost_minmax(gh[1], gh[2])
            else:
                gh = np.zeros((c.ne_ghost, nd + 2))
                ctx.set_ghost_records(gh)
            ghosts.append(gh)
        for k, (c, ctx) in enumerate(zip(cases, ctxs)):
            (rank, send, _), = c.peers
            recv = [r for rk, _, r in cases[rank].peers if rk == k][0]
            send = np.ascontiguousarray(send, dtype=np.int32)
            if layout == "arrays":
                rows, mn, mx = np.zeros((len(send), nd)), np.zeros(len(send)), np.zeros(len(send))
                ctx.halo_pack(us[k], send, len(send), rows, mn, mx)
                ghosts[rank][0][recv], ghosts[rank][1][recv], ghosts[rank][2][recv] = rows, mn, mx
            else:
                rec = np.zeros((len(send), nd + 2))
                ctx.halo_pack_records(us[k], send, len(send), rec)
                ghosts[rank][recv] = rec
        for k, (c, ctx) in enumerate(zip(cases, ctxs)):
            ctx.setup(t)
            y = np.zeros_like(us[k])
            assert 0 < c.ne_halo < c.ne_owned
            if layout == "arrays":
                ctx.stage_fused(us[k], dt, y)
            else:
                ctx.stage_fused_range(us[k], dt, y, c.ne_halo, c.ne_owned, False)
                ctx.stage_fused_range(us[k], dt, y, 0, c.ne_halo, True)
            assert np.array_equal(y, y_g[c.owned_gid])
            ctx.close()
    cg.close()


def test_bounds_type_1_and_dt_control(lib):
    """-bt 1 -dtc 1 (remhos.cpp:1178-1197, 1968-1998; remhos_tools.cpp:381-430) through the stepper and the C++
    driver, fused and granular: same accepted steps, same repeated steps, same final dt and field as the oracle
    (which is pinned for these options by autotest/out_baseline.dat:203-210).  lo 4: with the mass-based average
    the LO update never leaves the bounds and the controller never repeats a step."""
    from remhos_amd.case import Case, RmhdResult, make_config
    from remhos_amd.stepper import Stepper

    mesh, rs, p, prob, dt, tf = "periodic-cube", 0, 2, 0, 0.06, 0.12
    r = Remhos(Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=dt, t_final=tf, lo=4, fct=2, bounds_type=1, dt_control=1))
    out = r.run()
    assert r.repeats == 3 and out["steps"] == 4
    # (the GPU suite runs all four combinations on a longer case; here: driver granular, stepper fused)
    cfg = make_config(mesh, rs, p, prob, dt, tf, lo_type=4, fused=0, bounds_type=1, dt_control=1)
    res = RmhdResult()
    assert lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0
    assert (res.steps, res.repeats) == (out["steps"], r.repeats)
    assert abs(res.dt - out["dt"]) < 1e-12 * out["dt"]
    assert abs(res.final_mass - out["mass"]) < 1e-13 and abs(res.max_value - out["max"]) < 1e-12
    # (the stepper's fused path under the same options: tests/test_gpu_dtc.py and tests/test_dist_gloo.py)
    # -dtc 1 without -bt 1 is refused like remhos.cpp:617-620
    res = RmhdResult()
    assert lib.rmhd_run(C.byref(make_config(mesh, rs, p, prob, dt, tf, lo_type=4, dt_control=1)), C.byref(res)) != 0
    assert b"requires -bt 1" in lib.rmhd_last_error()


@pytest.mark.parametrize("p", [2, 3])
def test_mass_completion_emulated(lib, p):
    """rmh_set_mass_completion under the host emulation: with the constant mode the element's mass rate sum m du equals
    the converged solve's for a solve capped at one PCG iteration and for the reference's rule (abs 1e-8,
    remhos_ho.cpp:79-80), stand-alone HO kernel and one-kernel stage; the Jacobi step moves the field towards the
    converged one.  (GPU twin: tests/test_gpu_mass_completion.py.)"""
    from remhos_amd.capi import Context

    cfg = Config(mesh="cube01_hex", rs=0, order=p, problem=10, dt=0.02, t_final=0.7, lo=5)
    r = Remhos(cfg)
    r.refine_steps = 2
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=p, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    u = perturbed(r.u)
    t = 0.5
    ctx.setup(t)
    m = np.zeros_like(u)
    ctx.compute_lumped_mass(t, m)
    res = {}
    rules = dict(conv=(1e-14, 0, 100, 0, 0), cap1=(0, 0, 1, 0, 0), cap1fix=(0, 0, 1, 0, 1), cap1all=(0, 0, 1, 1, 1), refall=(0, 1e-8, 100, 1, 1))
    for name, (rel, ab, it, jac, fix) in rules.items():
        ctx.set_mass_tol(rel, ab, it)
        ctx.set_mass_completion(jac, fix)
        du, y, d2 = np.zeros_like(u), np.zeros_like(u), np.zeros_like(u)
        ctx.ho_apply(u, du)
        ctx.stage_fused(u, cfg.dt, y, du=d2)
        res[name] = (du, (m * du).sum(axis=1), (m * d2).sum(axis=1))
    ref = res["conv"]
    sc = np.abs(ref[1]).max()
    for k in ("cap1fix", "cap1all", "refall"):
        assert np.abs(res[k][1] - ref[1]).max() < 1e-13 * sc and np.abs(res[k][2] - ref[2]).max() < 1e-13 * sc, k
    assert np.abs(res["cap1"][1] - ref[1]).max() > 1e-4 * sc
    assert np.abs(res["cap1all"][0] - ref[0]).max() < 0.5 * np.abs(res["cap1fix"][0] - ref[0]).max()
    ctx.close()


def test_extrema_tokens(lib):
    """rmh_stage_fused_chain: the element extrema of a stage's output are reused only for a caller that presents the
    token that stage returned; no token or a stale one costs a recomputation and never gives wrong bounds (the
    pointer-keyed cache this replaces could: a vector modified in place kept its stale extrema)."""
    from remhos_amd.capi import Context

    cfg = Config(mesh="cube01_hex", rs=0, order=2, problem=10, dt=0.02, t_final=0.7, lo=5)
    r = Remhos(cfg)
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=2, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    u = perturbed(r.u)
    ctx.setup(0.3)
    y1, y2, z2 = np.zeros_like(u), np.zeros_like(u), np.zeros_like(u)
    t1 = ctx.stage_fused(u, cfg.dt, y1)
    assert t1 != 0
    t2 = ctx.stage_fused(y1, cfg.dt, y2, token=t1)          # extrema reused
    assert t2 not in (0, t1)
    ctx.stage_fused(y1, cfg.dt, z2)                         # recomputed
    assert np.array_equal(y2, z2)
    ctx.stage_fused(y1, cfg.dt, z2, token=t1)               # a stale token (two stages old) is not honoured
    assert np.array_equal(y2, z2)
    # a vector modified in place: its owner presents no token -- the new extrema are used
    y1b = y1.copy()
    y1b[0, :] += 0.25
    ref = np.zeros_like(u)
    ctx.stage_fused(y1b, cfg.dt, ref)
    t1 = ctx.stage_fused(u, cfg.dt, y1)
    y1[0, :] += 0.25
    ctx.stage_fused(y1, cfg.dt, z2, token=0)
    assert np.array_equal(ref, z2)
    # any other entry point in between drops the token
    t1 = ctx.stage_fused(u, cfg.dt, y1)
    dh = np.zeros_like(u)
    ctx.ho_apply(u, dh)
    ctx.stage_fused(y1, cfg.dt, z2, token=t1)
    ctx.stage_fused(y1, cfg.dt, ref)
    assert np.array_equal(ref, z2)
    ctx.close()


@pytest.mark.parametrize("mesh,p,rs,chunks", [("periodic-cube", 4, 0, ((3, 0), (2, 1))), ("cube01_hex", 3, 1, ((1, 0),))])
def test_xcd_chunk_order_is_a_permutation_of_the_batches(lib, mesh, p, rs, chunks, monkeypatch):
    """The XCD-aware batch order of the stage kernel (ho_kernel2: blockIdx.x -> batch; HoArgs::xcd_chunk, chunks of a lattice
    layer -- or of 2^weave layers woven batch by batch, xcd_weave -- dealt round-robin to the 8 XCDs, the rest in contiguous eighths):
    every (chunk, weave) must visit every batch exactly once -- the one-kernel stage then gives the same numbers bit by bit as with contiguous eighths (RMH_XCD_CHUNK = 0; that
    order is the one every other test of this file runs against the oracle).  27 one-element and 10 seven-element batches: whole
    rounds plus a tail; the full-size twins are tests/test_gpu_tile_order.py."""
    from remhos_amd.capi import Context

    cfg = Config(mesh=mesh, rs=rs, order=p, problem=10, dt=0.01, t_final=0.7, lo=5)
    r = Remhos(cfg)
    r.refine_steps = 2
    x0, vel, nbr, st = layout_from_oracle(r)
    u = perturbed(r.u)
    out = {}
    for chunk in ((0, 0),) + chunks:
        monkeypatch.setenv("RMH_XCD_CHUNK", str(chunk[0]))
        monkeypatch.setenv("RMH_XCD_WEAVE", str(chunk[1]))
        monkeypatch.setenv("RMH_ALT_ORDER", "0" if chunk == (0, 0) else "1")
        ctx = Context(lib, order=p, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
        ctx.setup(0.3)
        y, du = np.full_like(u, np.nan), np.full_like(u, np.nan)
        ctx.stage_fused(u, cfg.dt, y, dt_rk=cfg.dt, du=du)
        # (a second stage: with RMH_ALT_ORDER every other launch walks the batches backwards)
        y2, du2 = np.full_like(u, np.nan), np.full_like(u, np.nan)
        ctx.stage_fused(y, cfg.dt, y2, dt_rk=cfg.dt, du=du2)
        ctx.close()
        out[chunk] = (np.concatenate([y, y2]), np.concatenate([du, du2]))
    ref = out[(0, 0)]
    assert np.isfinite(ref[0]).all() and np.isfinite(ref[1]).all()
    for chunk in chunks:
        assert np.array_equal(out[chunk][0], ref[0]) and np.array_equal(out[chunk][1], ref[1]), chunk
