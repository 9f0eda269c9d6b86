"""The in-library neighbour exchange (rmh_exchange_*, include/rmh.h) with its same-process transport: all blocks of
a box partition as contexts of ONE process, device copies instead of RCCL -- same plan, pack kernels, ghost records
and stream/event ordering as the RCCL path.  CPU: through the g++ emulation build of the kernel sources; the GPU
twin is tests/test_gpu_exchange.py."""
import numpy as np
import pytest

from remhos_amd.capi import RmhError


def run_blocks(lib, device, mesh, rs, p, prob, part, steps, lo=5, compact=None, extra=(0, 0, 0)):
    """final field in global element order, from a lockstep run of all blocks in this process"""
    import os

    import torch

    from remhos_amd.case import Case, make_config
    from remhos_amd.stepper import Stepper, lockstep_step

    n = part[0] * part[1] * part[2]
    old = os.environ.get("RMH_COMPACT")
    if compact is not None:
        os.environ["RMH_COMPACT"] = "1" if compact else "0"
    try:
        cases = [Case(lib, make_config(mesh, rs, p, prob, -1.0, 0.5, lo_type=lo, part=part, rank=r, rs_extra=extra)) for r in range(n)]
        sts = [Stepper(lib, c, device=device) for c in cases]
    finally:
        if old is None:
            os.environ.pop("RMH_COMPACT", None)
        else:
            os.environ["RMH_COMPACT"] = old
    for s in sts:
        if s.case.peers:
            s.connect_local_peers(sts)
    for _ in range(steps):
        lockstep_step(sts, cases[0].dt)
    if device != "cpu":
        torch.cuda.synchronize()
    gid = np.concatenate([c.owned_gid for c in cases])
    u = np.concatenate([s.x.cpu().numpy() for s in sts])
    compact_used = [s.compact for s in sts]
    for s in sts:
        s.close()
    return u[np.argsort(gid)], compact_used


@pytest.fixture(scope="module")
def lib():
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver
    from tests.helpers import emu_library_path

    return bind_driver(load_library(emu_library_path()))


@pytest.mark.parametrize("mesh,rs,p,part,lo,compact", [
    ("cube01_hex", 1, 2, (2, 1, 1), 5, True),     # 4^3 elements, blocks 2 thick: face layers + extrema only
    ("cube01_hex", 1, 2, (2, 1, 1), 5, False),    # whole neighbour elements
    ("cube01_hex", 1, 1, (2, 2, 1), 5, True),     # edge neighbours: extrema-only records
    ("cube01_hex", 1, 2, (1, 2, 1), 4, True),     # subcell RD reads the same ghost traces
])
def test_blocks_in_one_process_equal_single_block(lib, mesh, rs, p, part, lo, compact):
    u1, _ = run_blocks(lib, "cpu", mesh, rs, p, 10, (1, 1, 1), 1, lo=lo)
    un, used = run_blocks(lib, "cpu", mesh, rs, p, 10, part, 1, lo=lo, compact=compact)
    assert all(c == compact for c in used)
    assert np.array_equal(u1, un)  # the partition and the record format may not change a single bit


def test_compact_records_refuse_thin_blocks(lib):
    """an element adjacent to one neighbour rank through two faces (periodic, 2 blocks, one of them 1 element thick)"""
    from remhos_amd.capi import Context
    from remhos_amd.case import Case, make_config

    case = Case(lib, make_config("periodic-cube", 0, 2, 10, -1.0, 0.5, part=(1, 2, 1), rank=0))
    ctx = Context(lib, order=2, exec_mode=1, x0=case.x0, vel=case.vel, face_nbr=case.face_nbr, stencil27=case.stencil27,
                  ne_ghost=case.ne_ghost)
    with pytest.raises(RmhError, match="compact = 0"):
        ctx.exchange_setup(case.peers, compact=True)
    ctx.exchange_setup(case.peers, compact=False)  # the full-record plan is accepted afterwards
    ctx.close()


def test_exchange_call_order_is_checked(lib):
    import torch

    from remhos_amd.capi import Context
    from remhos_amd.case import Case, make_config

    case = Case(lib, make_config("periodic-cube", 1, 1, 10, -1.0, 0.5, part=(2, 1, 1), rank=0))
    ctx = Context(lib, order=1, exec_mode=1, x0=case.x0, vel=case.vel, face_nbr=case.face_nbr, stencil27=case.stencil27,
                  ne_ghost=case.ne_ghost)
    u = torch.from_numpy(case.u0.copy())
    with pytest.raises(RmhError, match="no exchange plan"):
        ctx.exchange_begin(u)
    ctx.exchange_setup(case.peers, compact=True)
    with pytest.raises(RmhError, match="without rmh_exchange_begin"):
        ctx.exchange_end()
    ctx.exchange_begin(u)
    with pytest.raises(RmhError, match="not ended"):
        ctx.exchange_begin(u)
    with pytest.raises(RmhError, match="already set up"):
        ctx.exchange_setup(case.peers, compact=True)
    # segments: face ghosts carry 2 + D^2 doubles, the plan is symmetric for this 2-block periodic case
    sp, sn, gp, gn = ctx.exchange_buffers()
    rank, so, scount, ro, rcount = ctx.exchange_peer(0)
    assert rank == 1 and so == 0 and ro == 0 and scount == sn and rcount == gn
    D2 = 4
    nface = 2 * 6 * 6  # both x-faces of a 3 x 6 x 6 block face the other rank
    assert sn == nface * (2 + D2) and gn == sn
    ctx.close()


def test_partitioned_cpp_driver_equals_single_block(lib):
    """rmhd_run_partitioned (C++ time loop over all blocks of this process, in-library exchange, compact records)
    against rmhd_run on the undivided mesh: same steps, same max, mass equal up to the order of the block sums"""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    one, many = RmhdResult(), RmhdResult()
    cfg = make_config("cube01_hex", 1, 1, 10, -1.0, 0.5, max_steps=1)
    assert lib.rmhd_run(C.byref(cfg), C.byref(one)) == 0, lib.rmhd_last_error()
    cfgp = make_config("cube01_hex", 1, 1, 10, -1.0, 0.5, max_steps=1, part=(2, 1, 2))
    assert lib.rmhd_run_partitioned(C.byref(cfgp), None, 0, C.byref(many)) == 0, lib.rmhd_last_error()
    assert (many.steps, many.stages, many.global_dofs) == (one.steps, one.stages, one.global_dofs)
    assert many.max_value == one.max_value
    assert abs(many.final_mass - one.final_mass) < 1e-14 and abs(many.mass0 - one.mass0) < 1e-14


def minmax_exchange_blocks(lib, device, mesh, rs, p, part, compact):
    """rmh_exchange_minmax_* between the blocks of a partition held by this process: given element extrema (a function of
    the global element id, with some (+inf, -inf) pairs like the masked extrema of product remap) travel to the neighbours'
    ghost slots; rmh_bounds on every block must then equal rmh_bounds on the undivided mesh, element for element."""
    import torch

    from remhos_amd.capi import Context
    from remhos_amd.case import Case, make_config

    dev = torch.device(device)

    def extrema(gid):
        lo = np.sin(0.37 * gid) - 1.5
        hi = lo + 1.0 + np.cos(0.11 * gid) ** 2
        empty = gid % 7 == 3  # inactive elements: the identities of the reduction
        return np.where(empty, np.inf, lo), np.where(empty, -np.inf, hi)

    def mk(c):
        return Context(lib, order=p, exec_mode=c.exec_mode, x0=c.x0, vel=c.vel, face_nbr=c.face_nbr, stencil27=c.stencil27,
                       ne_ghost=c.ne_ghost, device=dev.index or 0)

    def to(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    g = Case(lib, make_config(mesh, rs, p, 10, -1.0, 0.5))
    cg = mk(g)
    lo, hi = extrema(g.owned_gid.astype(np.float64))
    umin_g, umax_g = to(np.zeros((g.ne_owned, g.ndof))), to(np.zeros((g.ne_owned, g.ndof)))
    cg.bounds(to(lo), to(hi), umin_g, umax_g)
    ref = {int(gid): (umin_g[k].cpu().numpy(), umax_g[k].cpu().numpy()) for k, gid in enumerate(g.owned_gid)}
    cg.close()
    n = part[0] * part[1] * part[2]
    cases = [Case(lib, make_config(mesh, rs, p, 10, -1.0, 0.5, part=part, rank=r)) for r in range(n)]
    ctxs = [mk(c) for c in cases]
    for c, ctx in zip(cases, ctxs):
        ctx.exchange_setup(c.peers, compact=compact)
    for r, (c, ctx) in enumerate(zip(cases, ctxs)):
        for k, (rank, _, _) in enumerate(c.peers):
            if rank > r:
                kk = [j for j, (r2, _, _) in enumerate(cases[rank].peers) if r2 == r][0]
                ctx.comm_connect_local(k, ctxs[rank], kk)
    xs = [tuple(to(a) for a in extrema(c.owned_gid.astype(np.float64))) for c in cases]
    for ctx, (a, b) in zip(ctxs, xs):
        ctx._check(ctx.lib.rmh_exchange_minmax_begin(ctx.h, a.data_ptr(), b.data_ptr()))
    for ctx in ctxs:
        ctx._check(ctx.lib.rmh_exchange_minmax_end(ctx.h))
    for c, ctx, (a, b) in zip(cases, ctxs, xs):
        umin, umax = to(np.zeros((c.ne_owned, c.ndof))), to(np.zeros((c.ne_owned, c.ndof)))
        ctx.bounds(a, b, umin, umax)
        umin, umax = umin.cpu().numpy(), umax.cpu().numpy()
        for k, gid in enumerate(c.owned_gid):
            assert np.array_equal(umin[k], ref[int(gid)][0]) and np.array_equal(umax[k], ref[int(gid)][1])
        ctx.close()


@pytest.mark.parametrize("mesh,rs,p,part,compact", [("cube01_hex", 1, 1, (2, 2, 1), True), ("periodic-cube", 0, 2, (1, 1, 3), False)])
def test_minmax_exchange_between_blocks(lib, mesh, rs, p, part, compact):
    minmax_exchange_blocks(lib, "cpu", mesh, rs, p, part, compact)
