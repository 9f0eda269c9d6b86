"""GPU part of tests/test_selfloop.py: the self-wrapped block exchanged through a ONE-RANK RCCL COMMUNICATOR -- the
grouped ncclSend / ncclRecv of rmh_exchange_begin, the dlopen'd RCCL table, the exchange stream and its event ordering
against the interior / halo launches, and rmh_allreduce run on the hardware at hand (1-GPU boxes).  Reference call
sites replaced: remhos_ho.cpp:122, remhos_tools.cpp:449-466, remhos.cpp:1412-1421."""
import ctypes as C

import numpy as np
import pytest

from tests.test_selfloop import plain_run, selfloop_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import torch

    assert torch.cuda.is_available()
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver

    return bind_driver(load_library())


@pytest.mark.parametrize("rs,p,wrap,lo,compact,prob,steps", [
    (2, 3, 1, 5, True, 10, 4),    # 12^3 elements: face layers + extrema records
    (2, 3, 2, 5, False, 10, 4),   # whole-element records
    (1, 3, 3, 4, True, 10, 3),    # subcell RD reads the same ghost traces
    (1, 4, 1, 5, True, 10, 2),    # generic (non wave-aligned) reductions
    (1, 6, 2, 5, True, 10, 2),
    (2, 3, 1, 5, True, 0, 4),     # transport: real upwind fluxes across the seam
])
def test_rccl_selfloop_equals_plain_periodic(lib, rs, p, wrap, lo, compact, prob, steps):
    u0 = plain_run(lib, "cuda:0", rs, p, steps, lo=lo, prob=prob)
    u1, tr, red = selfloop_run(lib, "cuda:0", rs, p, wrap, steps, lo=lo, compact=compact, prob=prob)
    assert tr == "rccl"
    assert red == [1.5, -2.0]  # rmh_allreduce over the one rank
    assert np.array_equal(u0, u1)


@pytest.mark.parametrize("first", ["0", "1"])
def test_rccl_selfloop_with_and_without_the_comm_first_hold(lib, first, monkeypatch):
    """RMH_COMM_FIRST: the interior launch held until the exchange stream has passed its wait for the pack kernel (default for
    blocks of >= 16 M dofs only, so the other tests of this file run without it): same field either way."""
    monkeypatch.setenv("RMH_COMM_FIRST", first)
    u0 = plain_run(lib, "cuda:0", 2, 3, 4, lo=5, prob=10)
    u1, tr, _ = selfloop_run(lib, "cuda:0", 2, 3, 1, 4, lo=5, compact=True, prob=10)
    assert tr == "rccl" and np.array_equal(u0, u1)


def test_rccl_selfloop_cpp_driver(lib, tmp_path):
    """rmhd_run_partitioned in its RCCL mode (unique id through the file, communicator of one rank, C++ stage loop on a
    non-default stream with interior / halo ranges, reductions through rmh_allreduce) against rmhd_run."""
    from remhos_amd.case import RmhdResult, make_config

    one, loop = RmhdResult(), RmhdResult()
    kw = dict(mesh="periodic-cube", rs=2, order=3, problem=10, dt=-1.0, t_final=0.5, max_steps=5, pa=1)
    cfg = make_config(**kw)
    assert lib.rmhd_run(C.byref(cfg), C.byref(one)) == 0, lib.rmhd_last_error()
    idfile = str(tmp_path / "rmh.id").encode()
    for attempt in range(2):  # the second run finds no stale record: rank 0 removed the file
        cfgp = make_config(self_wrap=1, warmup_steps=2, **kw)
        assert lib.rmhd_run_partitioned(C.byref(cfgp), idfile, 0, C.byref(loop)) == 0, lib.rmhd_last_error()
        assert not (tmp_path / "rmh.id").exists()
        assert (loop.steps, loop.stages, loop.global_dofs) == (one.steps, one.stages, one.global_dofs)
        assert loop.timed_stages == 3 * (5 - 2) and loop.transport == 1 and loop.n_peers == 1
        assert loop.comm_ranks == 1  # ncclCommCount of the library's communicator (rmh_comm_count)
        assert loop.t_rhs > 0 and loop.fom_rhs > 0  # sampled TimingData events, scaled to all timed steps
        assert loop.send_bytes_per_stage == loop.recv_bytes_per_stage > 0
        assert loop.max_value == one.max_value
        assert abs(loop.final_mass - one.final_mass) < 1e-13 and abs(loop.mass0 - one.mass0) < 1e-13  # (host sums in different element orders)
        assert loop.fom_wall > 0


def test_ghost_setters_refused_after_exchange_setup(lib):
    """rmh_set_ghost_* after rmh_exchange_setup would leave re-indexed tables pointing into a caller buffer"""
    import torch

    from remhos_amd.capi import RmhError
    from remhos_amd.case import Case, make_config
    from remhos_amd.stepper import Stepper

    case = Case(lib, make_config("periodic-cube", 1, 2, 10, -1.0, 0.5, self_wrap=1))
    st = Stepper(lib, case, device="cuda:0")
    buf = torch.zeros(case.ne_ghost * (case.ndof + 2), dtype=torch.float64, device="cuda:0")
    for call in (lambda: st.ctx.set_ghost_u(buf), lambda: st.ctx.set_ghost_minmax(buf, buf), lambda: st.ctx.set_ghost_records(buf)):
        with pytest.raises(RmhError):
            call()
    st.close()


@pytest.mark.parametrize("kw", [
    dict(mesh="periodic-cube", rs=1, order=3, problem=0, dt=0.01, t_final=0.5, max_steps=4, fused=0),
    dict(mesh="periodic-cube", rs=1, order=3, problem=10, dt=0.02, t_final=0.5, max_steps=3, fused=1, ps=1, ode_solver=13),
    dict(mesh="periodic-cube", rs=1, order=2, problem=10, dt=0.02, t_final=0.5, max_steps=3, fused=0, ps=1, ode_solver=12, lo_type=5),
    dict(mesh="periodic-cube", rs=1, order=4, problem=10, dt=0.02, t_final=0.5, max_steps=2, fused=0, lo_type=4),
], ids=["sequence-transport", "product-idp3-fused", "product-idp2-sequence", "sequence-lo4"])
def test_rccl_selfloop_solver_classes(lib, kw):
    """The solver classes over the neighbour exchange (one-rank RCCL communicator): ExchangeFaceNbrData of u and of us in
    the HO solver (remhos_ho.cpp:122), the min / max reduction of ComputeBounds for u and for the MASKED extrema of
    s = us / u (rmh_exchange_minmax_*; remhos_tools.cpp:461-466, remhos.cpp:1883-1886) -- product remap across rank
    boundaries.  Bit-identical to the plain periodic run."""
    from tests.test_selfloop import driver_fields

    u0, us0, r0 = driver_fields(lib, 0, **kw)
    for wrap in (1, 3):
        u1, us1, r1 = driver_fields(lib, wrap, **kw)
        assert np.array_equal(u0, u1) and np.array_equal(us0, us1)
        assert r0.max_value == r1.max_value
        if kw.get("ps"):
            assert r0.s_max == r1.s_max and abs(r0.final_mass_us - r1.final_mass_us) <= 1e-14 * abs(r0.final_mass_us)


def test_rccl_selfloop_rank_driver_product_remap(lib, tmp_path):
    """rmhd_run_rank: the solver-class loop of ONE BLOCK per process over RCCL (what `mpirun -np N ./remhos -ps -s 13` is in
    the reference), here the one-rank partition of a self-wrapped block: unique id through the file, exchanges of u, us and
    the masked extrema of s inside the classes, the report reduced with rmh_allreduce -- equal to the plain periodic run."""
    from remhos_amd.case import Case, RmhdResult, make_config

    kw = dict(mesh="periodic-cube", rs=1, order=3, problem=10, dt=0.02, t_final=0.5, max_steps=3, fused=1, ps=1, ode_solver=13, pa=1)
    plain, rank = RmhdResult(), RmhdResult()
    cfg0 = make_config(**kw)
    n = Case(lib, cfg0).u0.size
    u0, us0, u1, us1 = (np.zeros(n) for _ in range(4))
    assert lib.rmhd_run_state(C.byref(cfg0), C.byref(plain), u0.ctypes.data, us0.ctypes.data) == 0, lib.rmhd_last_error()
    cfg1 = make_config(self_wrap=2, **kw)
    case1 = Case(lib, cfg1)
    idfile = str(tmp_path / "rank.id").encode()
    assert lib.rmhd_run_rank(C.byref(cfg1), idfile, 0, C.byref(rank), u1.ctypes.data, us1.ctypes.data) == 0, lib.rmhd_last_error()
    assert not (tmp_path / "rank.id").exists()
    order = np.argsort(case1.owned_gid)
    nd = case1.ndof
    assert np.array_equal(u0.reshape(-1, nd), u1.reshape(-1, nd)[order]) and np.array_equal(us0.reshape(-1, nd), us1.reshape(-1, nd)[order])
    assert rank.max_value == plain.max_value and rank.s_max == plain.s_max
    assert abs(rank.final_mass - plain.final_mass) < 1e-14 and abs(rank.final_mass_us - plain.final_mass_us) < 1e-14
    assert abs(rank.mass0_us - plain.mass0_us) < 1e-14
