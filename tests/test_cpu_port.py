"""The C++/OpenMP CPU port (oracle/cpu_port.cpp; bench.py's cpu_baseline) against the numpy oracle
and the reference's ctest masses (remhos_tests.cpp:63-68, 81-86)."""
import numpy as np
import pytest

from oracle.cpu_port import CpuPort
from oracle.remhos_oracle import Config, Remhos
from tests.helpers import layout_from_oracle, perturbed


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("mesh,rs,p,prob,t", [("cube01_hex", 1, 2, 10, 0.3), ("periodic-cube", 0, 3, 10, 0.4),
                                             ("periodic-cube", 0, 2, 0, 0.0), ("cube01_hex", 0, 4, 10, 0.5)])
def test_stage_vs_oracle(mesh, rs, p, prob, t):
    cfg = Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=0.01, t_final=0.7, lo=5)
    r = Remhos(cfg)
    r.refine_steps = 2
    x0, vel, nbr, st = layout_from_oracle(r)
    u = perturbed(r.u)
    keep = {}
    r.stage(u, t, cfg.dt, keep)
    c = CpuPort(p, r.exec_mode, x0, vel, nbr, st, r.u)
    du, m, dh = c.stage(u, t, cfg.dt)
    assert _rel(m, keep["m"]) < 1e-13
    assert _rel(dh, keep["du_ho"]) < 1e-9
    assert _rel(du, keep["du"]) < 1e-9


@pytest.mark.parametrize("kw,ref,rtol", [
    (dict(mesh="cube01_hex", rs=1, order=2, problem=10, dt=-1.0, t_final=0.5, lo=5, max_steps=5), 0.11972857593296446, 1e-14),
    (dict(mesh="cube01_hex", rs=3, order=3, problem=10, dt=-1.0, t_final=0.5, lo=5, max_steps=1), 0.11601536511552431, 5e-13),
])
def test_reference_ctest_masses(kw, ref, rtol):
    r = Remhos(Config(**kw))
    x0, vel, nbr, st = layout_from_oracle(r)
    c = CpuPort(kw["order"], 1, x0, vel, nbr, st, r.u)
    for _ in range(kw["max_steps"]):
        c.step(r.dt)
    assert abs(c.mass(c.t) - ref) <= rtol * (1 + abs(ref))
