"""rmhd_config.tile_rows on the MI355X: the stage kernels give the same field bit for bit whatever the element numbering
(tests/test_tile_order.py has the host side and the emulated kernels), also through the C++ loops and with the halo first."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import torch

    assert torch.cuda.is_available()
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver

    return bind_driver(load_library())


@pytest.mark.parametrize("mesh,rs,p,prob,lo,T", [("periodic-cube", 2, 3, 10, 5, 4), ("periodic-cube", 1, 6, 10, 5, 2), ("cube01_hex", 2, 4, 10, 5, 3),
                                                 ("periodic-cube", 2, 3, 10, 4, 4), ("periodic-cube", 2, 2, 0, 5, 5), ("cube01_hex", 2, 3, 10, 3, 2)])
def test_tiled_run_is_bit_identical(lib, mesh, rs, p, prob, lo, T):
    from tests.test_tile_order import run_steps

    u0, _ = run_steps(lib, "cuda:0", mesh, rs, p, prob, lo, 0, 3)
    u1, _ = run_steps(lib, "cuda:0", mesh, rs, p, prob, lo, T, 3)
    assert np.array_equal(u0, u1)


def test_tiled_selfloop_keeps_halo_first(lib):
    from tests.test_tile_order import run_steps

    u0, nh0 = run_steps(lib, "cuda:0", "periodic-cube", 2, 3, 10, 5, 0, 2, self_wrap=1)
    u1, nh1 = run_steps(lib, "cuda:0", "periodic-cube", 2, 3, 10, 5, 4, 2, self_wrap=1)
    assert nh0 == nh1 > 0 and np.array_equal(u0, u1)


@pytest.mark.parametrize("fused", [1, 0])
def test_tiled_driver_run_reports_the_same_numbers(lib, fused):
    """rmhd_run (the solver classes / the one-kernel stage) on the tiled numbering: same final mass and maximum"""
    from remhos_amd.case import RmhdResult, make_config

    out = []
    for T in (0, 4):
        cfg = make_config("cube01_hex", 2, 3, 10, -1.0, 0.5, max_steps=4, fused=fused, pa=1, tile_rows=T)
        res = RmhdResult()
        assert lib.rmhd_run(C.byref(cfg), C.byref(res)) == 0, lib.rmhd_last_error()
        out.append((res.final_mass, res.max_value, res.steps))
    assert out[0][2] == out[1][2] == 4
    assert abs(out[0][0] - out[1][0]) <= 2e-15 * abs(out[0][0]) and out[0][1] == out[1][1]  # (the mass is a sum over elements: order of the terms)


@pytest.mark.parametrize("mesh,rs,p,lo,self_wrap", [("periodic-cube", 4, 3, 5, 0), ("periodic-cube", 4, 3, 4, 0), ("periodic-cube", 3, 6, 5, 0),
                                                   ("cube01_hex", 5, 4, 5, 0), ("periodic-cube", 4, 3, 5, 1)])
def test_xcd_batch_order_is_bit_identical(lib, mesh, rs, p, lo, self_wrap, monkeypatch):
    """The stage kernel's blockIdx -> batch map (HoArgs::xcd_chunk / xcd_weave: lattice layers woven and dealt round-robin to the
    XCDs, chosen by xcd_chunk_for from the element numbering) against contiguous eighths (RMH_XCD_CHUNK=0) and against other
    weaves, and with / without every other stage walking the batches backwards (RMH_ALT_ORDER), on meshes large enough for the
    layered order to be chosen (>= 256 batches per layer): same field bit for bit."""
    from tests.test_tile_order import run_steps

    out = []
    for env in ({"RMH_XCD_CHUNK": "0", "RMH_ALT_ORDER": "0"}, {}, {"RMH_XCD_WEAVE": "0"}, {"RMH_XCD_WEAVE": "2", "RMH_ALT_ORDER": "0"}):
        for k in ("RMH_XCD_CHUNK", "RMH_XCD_WEAVE", "RMH_ALT_ORDER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        u, _ = run_steps(lib, "cuda:0", mesh, rs, p, 10, lo, 0, 2, self_wrap=self_wrap)
        assert np.isfinite(u).all()
        out.append(u)
    for u in out[1:]:
        assert np.array_equal(out[0], u)


@pytest.mark.parametrize("mesh,rs,p,expect", [("periodic-cube", 5, 3, (9216, 7, 2632, 1)), ("periodic-cube", 4, 6, (2304, 1, 4608, 1)),
                                              ("cube01_hex", 5, 4, (4096, 1, 8192, 1))])
def test_batch_order_of_the_bench_meshes(lib, mesh, rs, p, expect):
    """the XCD-aware batch order bench.py's headline, p6 and cube01_p4 blocks run with (profiles/r05_xcd_layers.txt)"""
    from remhos_amd.capi import Context
    from remhos_amd.case import Case, make_config

    case = Case(lib, make_config(mesh, rs, p, 10, -1.0, 0.5, pa=1))
    ctx = Context(lib, order=p, exec_mode=case.exec_mode, x0=case.x0, vel=case.vel, face_nbr=case.face_nbr, stencil27=case.stencil27)
    try:
        assert ctx.batch_order(case.ne_owned) == expect
    finally:
        ctx.close()
