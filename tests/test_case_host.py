"""Host logic (C++ case builder, box partition, halo lists) against the oracle's lattice."""
import itertools

import numpy as np
import pytest

from oracle.remhos_oracle import Config, Remhos
from remhos_amd.case import Case, load_host_library, make_config
from tests.helpers import layout_from_oracle


@pytest.fixture(scope="module")
def lib():
    return load_host_library()


CASES = [
    ("cube01_hex", 1, 2, 10, -1.0, 0.5),
    ("periodic-cube", 1, 3, 10, -1.0, 0.5),
    ("periodic-cube", 1, 2, 0, 0.015, 2.0),
    # dim = 2 (build_case_2d): the meshes of the reference's 2-D ctest / autotest entries
    ("inline-quad", 1, 2, 14, -1.0, 0.5),
    ("inline-quad", 2, 3, 14, 0.0015, 0.75),
    ("periodic-square", 1, 3, 5, 0.004, 0.8),
    ("periodic-square", 0, 2, 0, 0.01, 0.5),
]


@pytest.mark.parametrize("mesh,rs,p,prob,dt,tf", CASES)
def test_case_matches_oracle(lib, mesh, rs, p, prob, dt, tf):
    r = Remhos(Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=dt, t_final=tf, lo=4))
    x0, vel, nbr, st = layout_from_oracle(r)
    c = Case(lib, make_config(mesh, rs, p, prob, dt, tf, lo_type=4))
    assert c.dt == r.dt
    assert np.array_equal(c.x0, x0)
    assert np.abs(c.vel - vel).max() < 1e-15
    assert np.abs(c.u0 - r.u).max() < 1e-15
    assert np.array_equal(c.face_nbr, nbr)
    assert np.array_equal(c.stencil27, st)
    if r.exec_mode == 1:
        assert np.abs(c.subcell_vel - r.Vs.transpose(0, 2, 1)).max() < 1e-15
    else:
        assert np.abs(c.subcell_vel - r.vel(r.Xs0).transpose(0, 2, 1)).max() < 1e-15


@pytest.mark.parametrize("mesh,rs,part,extra", [("periodic-cube", 1, (2, 1, 1), (0, 0, 0)), ("periodic-cube", 1, (2, 2, 2), (0, 0, 0)),
                                                ("cube01_hex", 2, (2, 2, 1), (0, 0, 0)), ("cube01_hex", 1, (3, 1, 2), (0, 0, 0)),
                                                # bench.py's weak-scaling lattices: partitioned directions refined once more
                                                ("periodic-cube", 0, (2, 1, 1), (1, 0, 0)), ("periodic-cube", 0, (2, 2, 1), (1, 1, 0)),
                                                ("cube01_hex", 1, (1, 2, 2), (0, 1, 1))])
def test_partition_is_consistent(lib, mesh, rs, part, extra):
    """Union of the blocks = the global lattice; stencils agree through global ids; send and
    receive lists of neighbouring ranks match element for element."""
    nr = part[0] * part[1] * part[2]
    g = Case(lib, make_config(mesh, rs, 2, 10, -1.0, 0.5, rs_extra=extra))
    cases = [Case(lib, make_config(mesh, rs, 2, 10, -1.0, 0.5, part=part, rank=k, rs_extra=extra)) for k in range(nr)]
    n0 = Case(lib, make_config(mesh, rs, 2, 10, -1.0, 0.5)).n[0]
    assert g.n == [n0 << extra[0], n0 << extra[1], n0 << extra[2]]
    if any(extra):
        # one -rs block per rank
        assert all(c.ne_owned == n0 ** 3 for c in cases)
    assert sum(c.ne_owned for c in cases) == g.ne_owned
    all_gid = np.concatenate([c.owned_gid for c in cases])
    assert np.array_equal(np.sort(all_gid), np.arange(g.ne_owned))
    for c in cases:
        assert c.dt == g.dt
        gid_of_local = np.concatenate([c.owned_gid, c.ghost_gid])
        # geometry and initial data are the rows of the global case
        assert np.array_equal(c.x0, g.x0[c.owned_gid])
        assert np.array_equal(c.vel, g.vel[c.owned_gid])
        assert np.array_equal(c.u0, g.u0[c.owned_gid])
        # stencil in global numbering equals the global stencil
        st = np.where(c.stencil27 >= 0, gid_of_local[np.maximum(c.stencil27, 0)], -1)
        assert np.array_equal(st, g.stencil27[c.owned_gid])
        fn = np.where(c.face_nbr >= 0, gid_of_local[np.maximum(c.face_nbr, 0)], -1)
        assert np.array_equal(fn, g.face_nbr[c.owned_gid])
        # halo-first order: exactly the first ne_halo elements reach a ghost, and only those are sent
        halo = (c.stencil27 >= c.ne_owned).any(axis=1)
        assert halo[:c.ne_halo].all() and not halo[c.ne_halo:].any()
        for _, send, _ in c.peers:
            assert (send < c.ne_halo).all()
        # every ghost is filled by exactly one peer
        filled = np.concatenate([r for _, _, r in c.peers]) if c.peers else np.zeros(0, np.int32)
        assert np.array_equal(np.sort(filled), np.arange(c.ne_ghost))
    for a, b in itertools.permutations(range(nr), 2):
        pa = {rk: (s, r) for rk, s, r in cases[a].peers}
        pb = {rk: (s, r) for rk, s, r in cases[b].peers}
        if b in pa:
            assert a in pb
            send_gid = cases[a].owned_gid[pa[b][0]]
            recv_gid = cases[b].ghost_gid[pb[a][1]]
            assert np.array_equal(send_gid, recv_gid)


def test_bad_config_is_reported(lib):
    with pytest.raises(RuntimeError, match="unknown lattice mesh"):
        Case(lib, make_config("star-q2", 1, 2, 10))
    with pytest.raises(RuntimeError, match="order"):
        Case(lib, make_config("cube01_hex", 1, 9, 10))


@pytest.mark.parametrize("mesh,rs,p,prob", [("periodic-cube", 0, 2, 0), ("cube01_hex", 1, 3, 10)])
def test_save_writes_mfem_mesh_and_gridfunction(lib, tmp_path, mesh, rs, p, prob):
    """-save: the MFEM mesh v1.0 / GridFunction text files parse back to the case's topology, node positions at
    pseudo-time t and field values (what remhos.cpp:1015-1030 writes with PrintAsOne / SaveAsOne)."""
    c = Case(lib, make_config(mesh, rs, p, prob, -1.0, 0.5))
    t = 0.25
    mp, gp = tmp_path / "meshHO.mesh", tmp_path / "sltn.gf"
    c.save(t, c.u0, mp, gp)
    tok = mp.read_text().split()
    assert tok[:3] == ["MFEM", "mesh", "v1.0"] and tok[tok.index("dimension") + 1] == "3"
    ie = tok.index("elements")
    ne = int(tok[ie + 1])
    assert ne == c.ne_owned
    el = np.array(tok[ie + 2: ie + 2 + 10 * ne], dtype=np.int64).reshape(ne, 10)
    assert (el[:, 1] == 5).all()  # Geometry::CUBE
    ib = tok.index("boundary")
    nb = int(tok[ib + 1])
    periodic = mesh.startswith("periodic")
    n1 = round(ne ** (1 / 3))
    assert nb == (0 if periodic else 6 * n1 * n1)
    iv = tok.index("vertices")
    nv = int(tok[iv + 1])
    assert nv == (n1 ** 3 if periodic else (n1 + 1) ** 3)
    assert el[:, 2:].min() == 0 and el[:, 2:].max() == nv - 1
    # each element has 8 distinct vertices; every vertex is used by 8 elements (periodic) / at most 8
    assert all(len(set(r)) == 8 for r in el[:, 2:])
    counts = np.bincount(el[:, 2:].ravel(), minlength=nv)
    assert counts.max() == 8 and (counts.min() == 8 if periodic else counts.min() == 1)
    inod = tok.index("nodes")
    assert tok[inod + 1: inod + 8] == ["FiniteElementSpace", "FiniteElementCollection:", "L2_T1_3D_P2", "VDim:", "3", "Ordering:", "0"]
    vals = np.array(tok[inod + 8:], dtype=np.float64).reshape(3, ne, 27)
    x_t = c.x0 + (t * c.vel if c.exec_mode == 1 else 0.0)  # [ne][3][27]
    order = np.argsort(c.owned_gid)
    assert np.allclose(vals, x_t[order].transpose(1, 0, 2), rtol=1e-13, atol=1e-13)
    g = gp.read_text().split()
    assert g[:7] == ["FiniteElementSpace", "FiniteElementCollection:", f"L2_T2_3D_P{p}", "VDim:", "1", "Ordering:", "0"]
    assert np.allclose(np.array(g[7:], dtype=np.float64).reshape(ne, -1), c.u0[order], rtol=1e-13, atol=1e-13)
    # multi-rank blocks are refused
    c2 = Case(lib, make_config(mesh, max(rs, 1), p, prob, -1.0, 0.5, part=(2, 1, 1), rank=0))
    with pytest.raises(RuntimeError, match="single-rank"):
        c2.save(t, None, tmp_path / "x.mesh")


@pytest.mark.parametrize("mesh", ["periodic-cube", "cube01_hex"])
def test_save_matches_reference_mesh_files(lib, mesh):
    """-save at -rs 0 against the mesh files the reference itself reads (data/periodic-cube.mesh, data/cube01_hex.mesh):
    same corner coordinates, same vertex sharing between elements (incl. the periodic identification), same section
    grammar and counts -- through numbering-independent canonical forms whose hashes tools/check_save_format.py derived
    from the reference files in the build container (tests/golden/save_format.json; only hashes and counts are stored)."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_save_format as csf

    want = json.load(open(os.path.join(root, "tests", "golden", "save_format.json")))[mesh]
    got = csf.summary(mesh, csf.parse_mesh(csf.our_file(mesh)))
    for k in ("elements", "vertices", "keywords", "geometry_sha256", "topology_sha256", "shared_vertex_pairs"):
        assert got[k] == want[k], k
