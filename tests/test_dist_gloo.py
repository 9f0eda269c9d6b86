"""N > 1 path on CPU: world_size-2 (and 4) gloo runs of the box-partitioned stepper must
reproduce the single-rank result -- same partition / halo-list / exchange code the GPU path uses
with the nccl (RCCL) backend.  The reference's suite checks the same rank-count invariance
(autotest/test.sh with mpirun -np N against one baseline)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(tmp_path, mesh, rs, p, prob, steps, part, opts=""):
    from tests.helpers import emu_library_path

    emu_library_path()  # (up to date before the workers load it)
    world = part[0] * part[1] * part[2]
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", RMH_TEST_OPTS=opts)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), mesh, str(rs), str(p),
             str(prob), str(steps)] + [str(k) for k in part], env=env))
    for pr in procs:
        assert pr.wait(timeout=900) == 0
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    gid = np.concatenate([d["gid"] for d in parts])
    u = np.concatenate([d["u"] for d in parts])
    order = np.argsort(gid)
    _run.info = [tuple(d["info"]) for d in parts]
    return u[order], float(parts[0]["mass"][0]), float(parts[0]["umax"][0])


@pytest.mark.parametrize("mesh,rs,p,prob,part", [("cube01_hex", 1, 1, 10, (2, 2, 1)),
                                                 # p = 3: the wavefront-aligned DPP reductions of the batched kernel, 3 ranks
                                                 ("periodic-cube", 0, 3, 10, (1, 3, 1)),
                                                 # 6^3 elements, 3x6x6 per rank: a non-empty interior range runs while
                                                 # the exchange is in flight (rmh_stage_fused_range), the shell after it
                                                 ("periodic-cube", 1, 1, 10, (2, 1, 1))])
def test_rank_count_invariance(tmp_path, mesh, rs, p, prob, part):
    from oracle.remhos_oracle import Config, Remhos

    steps = 1
    d1 = tmp_path / "n1"
    dn = tmp_path / "nn"
    d1.mkdir()
    dn.mkdir()
    u1, mass1, max1 = _run(d1, mesh, rs, p, prob, steps, (1, 1, 1))
    un, massn, maxn = _run(dn, mesh, rs, p, prob, steps, part)
    # identical arithmetic per element: the partition may not change a single bit of the field
    assert np.array_equal(u1, un)
    assert max1 == maxn
    assert abs(mass1 - massn) < 1e-15  # different summation order of the all-reduce
    # and both equal the oracle
    r = Remhos(Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=-1.0, t_final=0.5, lo=5, max_steps=steps))
    out = r.run()
    assert np.abs(un - r.u).max() < 1e-13
    assert abs(massn - out["mass"]) < 1e-14


def test_subcell_rd_and_dt_control_two_ranks(tmp_path):
    """-lo 4 (ghost traces in the lumped face fluxes of the RD solver) with -bt 1 -dtc 1 on two ranks: the
    controller sees the MIN over the ranks (all_reduce, MPI_Allreduce of remhos.cpp:1993), so both ranks repeat
    the same steps; field, step counts and the final dt equal the oracle's."""
    from oracle.remhos_oracle import Config, Remhos

    mesh, rs, p, prob, dt, tf = "periodic-cube", 0, 2, 0, 0.06, 0.12
    un, massn, maxn = _run(tmp_path, mesh, rs, p, prob, -1, (1, 2, 1), opts=f"4,1,1,{dt},{tf}")
    r = Remhos(Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=dt, t_final=tf, lo=4, fct=2, bounds_type=1, dt_control=1))
    out = r.run()
    assert r.repeats == 3
    for nsteps, repeats, dt_end in _run.info:
        assert (int(nsteps), int(repeats)) == (out["steps"], r.repeats)
        assert abs(dt_end - out["dt"]) < 1e-12 * out["dt"]
    assert np.abs(un - r.u).max() < 1e-12
    assert abs(massn - out["mass"]) < 1e-13
