"""GPU twin of tests/test_exchange_local.py: all blocks of a box partition as contexts on ONE MI355X, exchanged by the
library's same-process transport (device copies on the exchange streams, event-ordered against the contexts'
streams) -- the plan, pack kernels, compact ghost records and stream ordering of the RCCL path without a second GPU."""
import numpy as np
import pytest

from tests.test_exchange_local import run_blocks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver

    return bind_driver(load_library())


@pytest.mark.parametrize("mesh,rs,p,part,lo,compact,steps", [
    ("periodic-cube", 2, 3, (2, 2, 2), 5, True, 3),    # 12^3 elements, 8 blocks of 6^3: face layers + extrema
    ("periodic-cube", 2, 3, (2, 2, 2), 5, False, 3),   # whole neighbour elements
    ("periodic-cube", 1, 3, (2, 1, 1), 4, True, 2),    # subcell RD reads the same ghost traces
    ("cube01_hex", 2, 4, (2, 2, 1), 5, True, 2),       # bounded mesh, generic (non wave-aligned) reductions
    ("periodic-cube", 1, 6, (1, 1, 2), 5, True, 1),
])
def test_blocks_on_one_gpu_equal_single_block(lib, mesh, rs, p, part, lo, compact, steps):
    u1, _ = run_blocks(lib, "cuda:0", mesh, rs, p, 10, (1, 1, 1), steps, lo=lo)
    un, used = run_blocks(lib, "cuda:0", mesh, rs, p, 10, part, steps, lo=lo, compact=compact)
    assert all(c == compact for c in used)
    assert np.array_equal(u1, un)


def test_weak_scaling_lattice_blocks(lib):
    """bench.py's weak-scaling lattice for 2 ranks (x refined once more, one -rs block per rank) against the same
    lattice as one block"""
    u1, _ = run_blocks(lib, "cuda:0", "periodic-cube", 1, 3, 10, (1, 1, 1), 2, extra=(1, 0, 0))
    un, _ = run_blocks(lib, "cuda:0", "periodic-cube", 1, 3, 10, (2, 1, 1), 2, extra=(1, 0, 0))
    assert np.array_equal(u1, un)


def test_partitioned_cpp_driver_on_gpu(lib):
    """rmhd_run_partitioned: the C++ time loop over 8 blocks held by this process (in-library exchange with compact
    records, interior / halo ranges) against rmhd_run on the undivided mesh"""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    one, many = RmhdResult(), RmhdResult()
    cfg = make_config("periodic-cube", 2, 3, 10, -1.0, 0.5, max_steps=4)
    assert lib.rmhd_run(C.byref(cfg), C.byref(one)) == 0, lib.rmhd_last_error()
    cfgp = make_config("periodic-cube", 2, 3, 10, -1.0, 0.5, max_steps=4, part=(2, 2, 2))
    assert lib.rmhd_run_partitioned(C.byref(cfgp), None, 0, C.byref(many)) == 0, lib.rmhd_last_error()
    assert (many.steps, many.stages, many.global_dofs) == (one.steps, one.stages, one.global_dofs)
    assert many.max_value == one.max_value
    assert abs(many.final_mass - one.final_mass) < 1e-14 and abs(many.mass0 - one.mass0) < 1e-14
    # dt control: the min over the blocks drives the controller -- same accepted / repeated steps as one block
    kw = dict(bounds_type=1, dt_control=1, lo_type=4)
    cfg = make_config("periodic-cube", 0, 2, 0, 0.06, 0.12, **kw)
    cfgp = make_config("periodic-cube", 0, 2, 0, 0.06, 0.12, part=(1, 1, 3), **kw)
    assert lib.rmhd_run(C.byref(cfg), C.byref(one)) == 0, lib.rmhd_last_error()
    assert lib.rmhd_run_partitioned(C.byref(cfgp), None, 0, C.byref(many)) == 0, lib.rmhd_last_error()
    assert one.repeats > 0 and (many.steps, many.repeats, many.dt) == (one.steps, one.repeats, one.dt)
    assert many.max_value == one.max_value and abs(many.final_mass - one.final_mass) < 1e-14


@pytest.mark.parametrize("mesh,rs,p,part,compact", [("periodic-cube", 2, 3, (2, 2, 2), True), ("periodic-cube", 1, 4, (1, 2, 1), False),
                                                    ("cube01_hex", 2, 2, (2, 1, 2), True)])
def test_minmax_exchange_between_blocks_gpu(lib, mesh, rs, p, part, compact):
    """rmh_exchange_minmax_* (the masked extrema of product remap across ranks) with all blocks on one GPU"""
    from tests.test_exchange_local import minmax_exchange_blocks

    minmax_exchange_blocks(lib, "cuda:0", mesh, rs, p, part, compact)
