"""Element numbering of the case builder (rmhd_config.tile_rows): strips of T lattice rows in y, z before y inside a strip.
A numbering is not a different mesh -- MFEM's own element order after uniform refinement is no lattice order either, and
the reference's kernels are written per element (`forall(e, NE, ...)`, remhos_lo.cpp:1473-1612) -- so everything must come
out the same element for element: the case data through `owned_gid`, and whole RK steps bit for bit (kernel sources under
the host emulation here; on the MI355X in tests/test_gpu_tile_order.py)."""
import numpy as np
import pytest

from remhos_amd.case import Case, make_config


@pytest.fixture(scope="module")
def hostlib():
    from remhos_amd.case import load_host_library

    return load_host_library()


@pytest.fixture(scope="module")
def emulib():
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver
    from tests.helpers import emu_library_path

    return bind_driver(load_library(emu_library_path()))


@pytest.mark.parametrize("mesh,rs,p,prob,T", [("periodic-cube", 1, 2, 10, 2), ("cube01_hex", 2, 3, 10, 3), ("periodic-cube", 2, 1, 0, 4),
                                                 ("periodic-cube", 1, 2, 10, 100)])
def test_tiled_case_is_the_same_mesh(hostlib, mesh, rs, p, prob, T):
    a = Case(hostlib, make_config(mesh, rs, p, prob, -1.0 if prob >= 10 else 0.01, 0.5, lo_type=4))
    b = Case(hostlib, make_config(mesh, rs, p, prob, -1.0 if prob >= 10 else 0.01, 0.5, lo_type=4, tile_rows=T))
    assert a.ne_owned == b.ne_owned and a.dt == b.dt and b.ne_halo == 0
    assert np.array_equal(a.owned_gid, np.arange(a.ne_owned))  # lattice order is the global id
    n = round(a.ne_owned ** (1 / 3))
    # (one strip that holds every row is the lattice order again)
    assert sorted(b.owned_gid) == list(range(b.ne_owned)) and np.array_equal(b.owned_gid, a.owned_gid) == (T >= n)
    g = b.owned_gid
    for name in ("x0", "vel", "u0", "subcell_vel"):
        assert np.array_equal(getattr(b, name), getattr(a, name)[g]), name
    # neighbour tables: the same neighbours, named by their new numbers
    new_of_gid = np.empty(b.ne_owned, dtype=np.int64)
    new_of_gid[g] = np.arange(b.ne_owned)
    for name in ("face_nbr", "stencil27"):
        want = getattr(a, name)[g].copy()
        m = want >= 0
        want[m] = new_of_gid[want[m]]
        assert np.array_equal(getattr(b, name), want), name
    # the order itself: strips of T rows in y; inside a strip z runs before y, x fastest
    lx, ly, lz = g % n, (g // n) % n, g // (n * n)
    key = lx + n * ((ly % T) + T * (lz + n * (ly // T)))
    assert np.all(np.diff(key) > 0)


def run_steps(lib, device, mesh, rs, p, prob, lo, T, steps, part=(1, 1, 1), self_wrap=0):
    from remhos_amd.stepper import Stepper

    case = Case(lib, make_config(mesh, rs, p, prob, -1.0 if prob >= 10 else 0.01, 0.5, lo_type=lo, tile_rows=T, pa=1, self_wrap=self_wrap))
    st = Stepper(lib, case, device=device)
    for _ in range(steps):
        st.step(case.dt)
    if device != "cpu":
        import torch

        torch.cuda.synchronize()
    u = st.x.cpu().numpy()[np.argsort(case.owned_gid)]
    nh = case.ne_halo
    st.close()
    return u, nh


@pytest.mark.parametrize("mesh,rs,p,prob,lo,T", [("periodic-cube", 1, 2, 10, 5, 2), ("cube01_hex", 1, 3, 10, 4, 2)])  # (transport, p = 3 ... 6: tests/test_gpu_tile_order.py)
def test_tiled_run_is_bit_identical_emulated(emulib, mesh, rs, p, prob, lo, T):
    u0, _ = run_steps(emulib, "cpu", mesh, rs, p, prob, lo, 0, 1)
    u1, _ = run_steps(emulib, "cpu", mesh, rs, p, prob, lo, T, 1)
    assert np.array_equal(u0, u1)


def test_tiled_order_keeps_the_halo_first_emulated(emulib):
    """with ghosts (self-wrapped block: the exchange runs for real) the halo shell still comes first, the strips behind it"""
    u0, nh0 = run_steps(emulib, "cpu", "periodic-cube", 1, 2, 10, 5, 0, 1, self_wrap=1)
    u1, nh1 = run_steps(emulib, "cpu", "periodic-cube", 1, 2, 10, 5, 2, 1, self_wrap=1)
    assert nh0 == nh1 > 0 and np.array_equal(u0, u1)


@pytest.mark.parametrize("mesh,rs,p,lo,T,expect", [
    ("periodic-cube", 4, 3, 5, 0, (2304, 7, 658, 1)),   # 48^3, seven elements per workgroup: 329 batches per layer, two layers woven, three rounds
    ("periodic-cube", 3, 6, 5, 0, (576, 1, 1152, 1)),   # 24^3, one element per workgroup: one round of 8 chunks + a tail in contiguous eighths
    ("periodic-cube", 3, 4, 4, 0, (576, 1, 1152, 1)),   # the subcell-RD stage kernel batches alike at p >= 4
    ("periodic-cube", 4, 3, 5, 4, (192, 7, 0, 0)),      # y-strips of 4 rows: "layers" of 27 batches are too small to hold an XCD's workgroups
    ("periodic-cube", 1, 3, 5, 0, (36, 7, 0, 0)),       # 6^3: contiguous eighths
    ("cube01_hex", 3, 4, 5, 0, (256, 1, 512, 1)),       # 16^3 non-periodic
])
def test_batch_order_chosen_from_the_numbering(emulib, mesh, rs, p, lo, T, expect):
    """rmh_batch_order: the layer length read off face_nbr, and the chunk / weave of the XCD-aware batch order it leads to
    (xcd_chunk_for, remhos_amd/csrc/rmh_api.hip); the bench meshes' values are asserted on the GPU (tests/test_gpu_tile_order.py)."""
    from remhos_amd.capi import Context

    case = Case(emulib, make_config(mesh, rs, p, 10, -1.0, 0.5, lo_type=lo, tile_rows=T, pa=1))
    ctx = Context(emulib, order=p, exec_mode=case.exec_mode, x0=case.x0, vel=case.vel, face_nbr=case.face_nbr, stencil27=case.stencil27,
                  subcell_vel=case.subcell_vel)
    ctx.set_lo_type(lo)
    try:
        assert ctx.batch_order(case.ne_owned) == expect
        assert ctx.batch_order(0)[2] == 0 and ctx.batch_order(64)[2] == 0  # (short launches -- a halo shell -- keep contiguous eighths)
    finally:
        ctx.close()
