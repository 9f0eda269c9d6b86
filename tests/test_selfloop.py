"""Self-loop of the neighbour exchange on ONE rank (rmhd_config.self_wrap): the periodic wrap of one direction of a
1 x 1 x 1 block becomes a halo whose ghosts are owned by the rank itself, so that the whole exchange path -- plan, pack
kernels, compact / full ghost records, interior + halo launch ranges, event ordering, and on the GPU the grouped
ncclSend / ncclRecv of rmh_exchange_begin on a one-rank RCCL communicator -- runs for real on a 1-GPU box.  What it
replaces in the reference: ParGridFunction::ExchangeFaceNbrData (remhos_ho.cpp:122) + the GroupCommunicator min/max
of DofInfo::ComputeOverlapBounds (remhos_tools.cpp:449-466).  Expected: bit-identical to the plain periodic run.

CPU part: host case builder + the kernel sources under the host emulation with the same-process transport."""
import numpy as np
import pytest


def selfloop_run(lib, device, rs, p, wrap, steps, lo=5, compact=True, prob=10):
    """final field in global element order of the self-wrapped run, and the Stepper's transport"""
    import os

    from remhos_amd.case import Case, make_config
    from remhos_amd.stepper import Stepper

    old = os.environ.get("RMH_COMPACT")
    os.environ["RMH_COMPACT"] = "1" if compact else "0"
    try:
        case = Case(lib, make_config("periodic-cube", rs, p, prob, -1.0 if prob >= 10 else 0.01, 0.5, lo_type=lo, self_wrap=wrap))
        st = Stepper(lib, case, device=device)
    finally:
        if old is None:
            os.environ.pop("RMH_COMPACT", None)
        else:
            os.environ["RMH_COMPACT"] = old
    assert st.compact == compact
    for _ in range(steps):
        st.step(case.dt)
    if device != "cpu":
        import torch

        torch.cuda.synchronize()
    u = st.x.cpu().numpy()[np.argsort(case.owned_gid)]
    tr = st.transport
    red = st.ctx.allreduce([1.5, -2.0], "sum") if tr == "rccl" else None
    st.close()
    return u, tr, red


def plain_run(lib, device, rs, p, steps, lo=5, prob=10):
    from remhos_amd.case import Case, make_config
    from remhos_amd.stepper import Stepper

    case = Case(lib, make_config("periodic-cube", rs, p, prob, -1.0 if prob >= 10 else 0.01, 0.5, lo_type=lo))
    st = Stepper(lib, case, device=device)
    for _ in range(steps):
        st.step(case.dt)
    if device != "cpu":
        import torch

        torch.cuda.synchronize()
    u = st.x.cpu().numpy()[np.argsort(case.owned_gid)]
    st.close()
    return u


@pytest.fixture(scope="module")
def hostlib():
    from remhos_amd.case import load_host_library

    return load_host_library()


@pytest.mark.parametrize("wrap", [1, 2, 3])
def test_selfloop_case_lists(hostlib, wrap):
    """the builder's halo lists of a self-wrapped block: one peer = the rank itself, ghosts = the two element layers
    at the seam, sent in ghost order; everything else is the plain periodic block"""
    from remhos_amd.case import Case, make_config

    c = Case(hostlib, make_config("periodic-cube", 1, 2, 10, -1.0, 0.5, self_wrap=wrap))
    n = c.n
    d = wrap - 1
    layer = n[0] * n[1] * n[2] // n[d]
    assert c.ne_owned == n[0] * n[1] * n[2] and c.ne_ghost == 2 * layer
    assert len(c.peers) == 1 and c.peers[0][0] == 0
    _, send, recv = c.peers[0]
    assert len(send) == len(recv) == 2 * layer and (recv == np.arange(2 * layer)).all()
    # the k-th record sent is the element whose copy the k-th ghost slot holds
    assert (c.owned_gid[send] == c.ghost_gid).all()
    coord = (c.ghost_gid // (1 if d == 0 else (n[0] if d == 1 else n[0] * n[1]))) % n[d]
    assert set(coord.tolist()) == {0, n[d] - 1}
    # tables: ghosts are referenced exactly by the elements of the two seam layers, through the wrapped direction
    gh = c.face_nbr >= c.ne_owned
    assert gh.sum() == 2 * layer and not gh[:, [f for f in range(6) if f // 2 != d]].any()
    assert c.ne_halo == 2 * layer
    plain = Case(hostlib, make_config("periodic-cube", 1, 2, 10, -1.0, 0.5))
    order, porder = np.argsort(c.owned_gid), np.argsort(plain.owned_gid)
    assert np.array_equal(c.x0[order], plain.x0[porder]) and np.array_equal(c.u0[order], plain.u0[porder])
    with pytest.raises(RuntimeError):
        Case(hostlib, make_config("cube01_hex", 1, 2, 10, -1.0, 0.5, self_wrap=1))  # not periodic


@pytest.fixture(scope="module")
def emulib():
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver
    from tests.helpers import emu_library_path

    return bind_driver(load_library(emu_library_path()))


@pytest.mark.parametrize("p,wrap,lo,compact", [(2, 1, 5, True), (2, 3, 5, False), (2, 2, 4, True)])
def test_selfloop_emulated_equals_plain_periodic(emulib, p, wrap, lo, compact):
    u0 = plain_run(emulib, "cpu", 0, p, 1, lo=lo)
    u1, tr, _ = selfloop_run(emulib, "cpu", 0, p, wrap, 1, lo=lo, compact=compact)
    assert tr == "local"
    assert np.array_equal(u0, u1)


def driver_fields(lib, self_wrap, **kw):
    """final u (and us) of rmhd_run_state in GLOBAL element order"""
    import ctypes as C

    from remhos_amd.case import Case, RmhdResult, make_config

    cfg = make_config(self_wrap=self_wrap, **kw)
    case = Case(lib, cfg)
    n = case.u0.size
    u, us = np.zeros(n), np.zeros(n)
    res = RmhdResult()
    assert lib.rmhd_run_state(C.byref(cfg), C.byref(res), u.ctypes.data, us.ctypes.data) == 0, lib.rmhd_last_error()
    order = np.argsort(case.owned_gid)
    return u.reshape(case.u0.shape)[order], us.reshape(case.u0.shape)[order], res


SOLVER_CLASS_RUNS = [
    # the reference's call sequence (HOSolver / LOSolver / DofInfo / FCTSolver one by one): every exchange the reference's
    # classes make -- ExchangeFaceNbrData in the HO solver, the min / max reduction in ComputeBounds -- goes through the plan
    dict(mesh="periodic-cube", rs=0, order=2, problem=0, dt=0.02, t_final=0.5, max_steps=1, fused=0),
    # product remap with the IDP solver: exchanges of u, of us, and of the masked extrema of s = us / u
    # (forward Euler IDP here: one stage per step keeps the emulated run short; -s 13 / 12 run on the GPU)
    dict(mesh="periodic-cube", rs=0, order=2, problem=10, dt=0.02, t_final=0.5, max_steps=2, fused=1, ps=1, ode_solver=11),
]


@pytest.mark.parametrize("kw", SOLVER_CLASS_RUNS, ids=["sequence-transport", "product-idp1"])
def test_selfloop_solver_classes_emulated(emulib, kw, monkeypatch):
    monkeypatch.setenv("RMH_EXCHANGE", "local")
    u0, us0, r0 = driver_fields(emulib, 0, **kw)
    u1, us1, r1 = driver_fields(emulib, 2, **kw)
    assert np.array_equal(u0, u1) and np.array_equal(us0, us1)
    assert r0.max_value == r1.max_value and abs(r0.final_mass - r1.final_mass) <= 1e-14 * abs(r0.final_mass)


def test_state_checks_of_the_exchange_and_the_split_stage(emulib):
    """Round-4 state checks (advisor findings): (i) the ranges of one fused stage must name the same u and dt -- a stage
    that is abandoned or fails leaves no stale extrema behind; (ii) after rmh_exchange_minmax_* the ghost extrema belong to
    another field: the limiters of u refuse them -- for the element ranges that read ghosts -- until u is exchanged again; (iii) rmh_comm_count without a communicator."""
    import pytest as _pytest
    import torch

    from remhos_amd.case import Case, make_config
    from remhos_amd.stepper import Stepper

    case = Case(emulib, make_config("periodic-cube", 0, 2, 10, -1.0, 0.5, self_wrap=1))
    st = Stepper(emulib, case, device="cpu")
    c, u, dt = st.ctx, st.x, case.dt
    assert st.transport == "local" and c.comm_count() == 0
    y, z = torch.zeros_like(u), torch.zeros_like(u)
    ne, nh = case.ne_owned, case.ne_halo
    c.setup(0.1)
    # reference: the split stage as the stepper runs it
    c.exchange_begin(u)
    c.stage_fused_range(u, dt, y, nh, ne, False)
    c.exchange_end()
    c.stage_fused_range(u, dt, y, 0, nh, True)
    # (i) a second range with another input vector (or another dt) is refused, and the stage is forgotten ...
    u2 = u.clone()
    u2[0, :] += 0.125
    c.exchange_begin(u)
    c.stage_fused_range(u, dt, z, nh, ne, False)
    c.exchange_end()
    with _pytest.raises(RuntimeError, match="same u and dt"):
        c.stage_fused_range(u2, dt, z, 0, nh, True)
    with _pytest.raises(RuntimeError, match="same u and dt"):
        c.exchange_begin(u)
        c.stage_fused_range(u, dt, z, nh, ne, False)
        c.exchange_end()
        c.stage_fused_range(u, 0.5 * dt, z, 0, nh, True)
    # ... so that the next stage starts from scratch (its own extrema) and reproduces the reference
    c.exchange_begin(u)
    c.stage_fused_range(u, dt, z, nh, ne, False)
    c.exchange_end()
    c.stage_fused_range(u, dt, z, 0, nh, True)
    assert torch.equal(y, z)
    # (ii) caller-given extrema in the ghost slots: the limiter of u refuses them, an exchange of u makes them valid again
    xe_min, xe_max = torch.zeros(ne, dtype=u.dtype), torch.ones(ne, dtype=u.dtype)
    c.exchange_minmax(xe_min, xe_max)
    with _pytest.raises(RuntimeError, match="another field"):
        c.stage_fused_range(u, dt, z, 0, nh, True)  # the halo shell reads the ghost extrema
    # (round 5, advisor: the interior range reads no ghost and is NOT refused; a range that does is refused while the
    # exchange is in flight -- the ghost extrema are valid from rmh_exchange_end on, not from rmh_exchange_begin)
    c.exchange_begin(u)
    c.stage_fused_range(u, dt, z, nh, ne, False)
    with _pytest.raises(RuntimeError, match="in flight"):
        c.stage_fused_range(u, dt, z, 0, nh, True)
    # (round 6, advisor: the granular ghost readers too -- rmh_ho_apply reads the neighbour traces, rmh_bounds the ghost extrema)
    with _pytest.raises(RuntimeError, match="in flight"):
        c.ho_apply(u, z)
    xi_lo, xi_hi = torch.zeros_like(u), torch.zeros_like(u)
    with _pytest.raises(RuntimeError, match="in flight"):
        c.bounds(xe_min, xe_max, xi_lo, xi_hi)
    c.exchange_end()
    c.ho_apply(u, z)  # valid ghosts: accepted; and rmh_bounds takes ANOTHER field's extrema from the ghost slots by design
    c.exchange_minmax(xe_min, xe_max)
    c.bounds(xe_min, xe_max, xi_lo, xi_hi)
    c.exchange_begin(u)
    c.stage_fused_range(u, dt, z, nh, ne, False)
    c.exchange_end()
    c.stage_fused_range(u, dt, z, 0, nh, True)
    assert torch.equal(y, z)
    st.close()
