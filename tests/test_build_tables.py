"""rmh_build_tables (include/rmh.h): face_nbr / stencil27 from the vertex ids of the elements, for ANY element
numbering -- the 'lattice order' assumption of the case builder is not part of the boundary.  The tables built from
a randomly renumbered lattice must equal the renumbered lattice tables, and a stage run on the renumbered mesh must
equal the original run element for element."""
import ctypes as C

import numpy as np
import pytest

from remhos_amd.case import Case, load_host_library, make_config


def lattice_vertices(n, periodic):
    """[ne][8] vertex ids of the n[0] x n[1] x n[2] lattice in lexicographic corner order, x fastest"""
    nx, ny, nz = n
    vx, vy, vz = (nx, ny, nz) if periodic else (nx + 1, ny + 1, nz + 1)
    ev = np.empty((nx * ny * nz, 8), dtype=np.int32)
    e = 0
    for ez in range(nz):
        for ey in range(ny):
            for ex in range(nx):
                for k in range(8):
                    ix, iy, iz = ex + (k & 1), ey + ((k >> 1) & 1), ez + (k >> 2)
                    if periodic:
                        ix, iy, iz = ix % nx, iy % ny, iz % nz
                    ev[e, k] = ix + vx * (iy + vy * iz)
                e += 1
    return ev


def build(lib, ne_owned, ev):
    ev = np.ascontiguousarray(ev, dtype=np.int32)
    nbr = np.empty((ne_owned, 6), dtype=np.int32)
    st = np.empty((ne_owned, 27), dtype=np.int32)
    rc = lib.rmh_build_tables(ne_owned, len(ev), ev.ctypes.data, nbr.ctypes.data, st.ctypes.data)
    return rc, nbr, st


@pytest.fixture(scope="module")
def lib():
    lib = load_host_library()
    lib.rmh_build_tables.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


@pytest.mark.parametrize("mesh,rs", [("periodic-cube", 0), ("periodic-cube", 1), ("cube01_hex", 1), ("cube01_hex", 2)])
def test_tables_from_vertices_equal_lattice_tables(lib, mesh, rs):
    c = Case(lib, make_config(mesh, rs, 2, 10, -1.0, 0.5))
    ev = lattice_vertices(c.n, mesh.startswith("periodic"))
    rc, nbr, st = build(lib, c.ne_owned, ev)
    assert rc == 0
    assert np.array_equal(nbr, c.face_nbr) and np.array_equal(st, c.stencil27)
    # any numbering: new index of element e is perm[e]
    rng = np.random.default_rng(7)
    perm = rng.permutation(c.ne_owned).astype(np.int32)
    inv = np.argsort(perm)
    vperm = rng.permutation(ev.max() + 1).astype(np.int32)  # vertex ids are arbitrary too
    rc, nbr_p, st_p = build(lib, c.ne_owned, vperm[ev][inv])
    assert rc == 0
    remap = lambda t: np.where(t >= 0, perm[np.maximum(t, 0)], -1)
    assert np.array_equal(nbr_p, remap(c.face_nbr)[inv]) and np.array_equal(st_p, remap(c.stencil27)[inv])


def test_ghost_elements_appear_as_entries_only(lib):
    """block of a partition: owned elements first, the ghost elements of the other ranks appended"""
    g = Case(lib, make_config("periodic-cube", 1, 2, 10, -1.0, 0.5))
    c = Case(lib, make_config("periodic-cube", 1, 2, 10, -1.0, 0.5, part=(2, 1, 1), rank=0))
    ev_g = lattice_vertices(g.n, True)
    ev = np.concatenate([ev_g[c.owned_gid], ev_g[c.ghost_gid]])
    rc, nbr, st = build(lib, c.ne_owned, ev)
    assert rc == 0
    assert np.array_equal(nbr, c.face_nbr) and np.array_equal(st, c.stencil27)


def test_misaligned_neighbour_is_refused(lib):
    ev = lattice_vertices((2, 1, 1), False)
    ev[1] = ev[1][[1, 3, 0, 2, 5, 7, 4, 6]]  # second element rotated about z
    rc, _, _ = build(lib, 2, ev)
    assert rc != 0


def test_stage_on_renumbered_mesh(lib):
    """end to end through the emulated kernels: same field, element for element, after a random renumbering"""
    import torch  # noqa: F401

    from remhos_amd.capi import Context, load_library
    from tests.helpers import emu_library_path

    emu = load_library(emu_library_path())
    c = Case(lib, make_config("periodic-cube", 0, 2, 10, -1.0, 0.5))
    ev = lattice_vertices(c.n, True)
    perm = np.random.default_rng(3).permutation(c.ne_owned).astype(np.int32)
    inv = np.argsort(perm)
    rc, nbr_p, st_p = build(lib, c.ne_owned, ev[inv])
    assert rc == 0

    def stage(x0, vel, nbr, st, u):
        ctx = Context(emu, order=2, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
        ctx.setup(0.3)
        y = np.zeros_like(u)
        ctx.stage_fused(np.ascontiguousarray(u), c.dt, y)
        ctx.close()
        return y

    y = stage(c.x0, c.vel, c.face_nbr, c.stencil27, c.u0)
    yp = stage(c.x0[inv], c.vel[inv], nbr_p, st_p, c.u0[inv])
    assert np.array_equal(yp, y[inv])


@pytest.mark.parametrize("mesh,rs", [("inline-quad", 0), ("inline-quad", 1), ("periodic-square", 0), ("periodic-square", 1)])
def test_tables_2d_from_vertices_equal_oracle_lattice_tables(lib, mesh, rs):
    """rmh_build_tables_2d: the quadrilateral analogue (4 corners, 4 faces, 3 x 3 stencil) against the oracle's lattice tables
    -- the inputs of the dim = 2 tests (tests/test_2d.py) -- also for a random element and vertex numbering."""
    from oracle.remhos_oracle import make_lattice
    from tests.helpers import layout_from_oracle

    lat = make_lattice(mesh, rs, 2)
    nx, ny = lat.n
    per = mesh.startswith("periodic")
    vx, vy = (nx, ny) if per else (nx + 1, ny + 1)
    ev = np.empty((nx * ny, 4), dtype=np.int32)
    for ey in range(ny):
        for ex in range(nx):
            for k in range(4):
                ix, iy = ex + (k & 1), ey + (k >> 1)
                if per:
                    ix, iy = ix % nx, iy % ny
                ev[ex + nx * ey, k] = ix + vx * iy

    class R:  # what layout_from_oracle reads for the tables
        pass

    nbr_o = lat.face_neighbors().astype(np.int32)
    st_o = np.stack([lat.shifted((ox, oy)) for oy in (-1, 0, 1) for ox in (-1, 0, 1)], axis=1).astype(np.int32)
    lib.rmh_build_tables_2d.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

    def build2(ev_):
        ev_ = np.ascontiguousarray(ev_, dtype=np.int32)
        nbr, st = np.empty((len(ev_), 4), dtype=np.int32), np.empty((len(ev_), 9), dtype=np.int32)
        return lib.rmh_build_tables_2d(len(ev_), len(ev_), ev_.ctypes.data, nbr.ctypes.data, st.ctypes.data), nbr, st

    rc, nbr, st = build2(ev)
    assert rc == 0 and np.array_equal(nbr, nbr_o) and np.array_equal(st, st_o)
    rng = np.random.default_rng(11)
    perm = rng.permutation(len(ev)).astype(np.int32)
    inv = np.argsort(perm)
    vperm = rng.permutation(ev.max() + 1).astype(np.int32)
    rc, nbr_p, st_p = build2(vperm[ev][inv])
    assert rc == 0
    remap = lambda t: np.where(t >= 0, perm[np.maximum(t, 0)], -1)  # noqa: E731
    assert np.array_equal(nbr_p[perm], remap(nbr_o)) and np.array_equal(st_p[perm], remap(st_o))
