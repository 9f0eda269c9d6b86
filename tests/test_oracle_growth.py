"""Why whole runs at p = 6 cannot be compared digit for digit at the reference's time step -- shown on the CPU oracle, with
no kernel involved.

`-dt -1` is the p-independent rule 0.25 h / |v| (remhos.cpp:538-553).  At p = 6 that step is far beyond the explicit
stability limit of the unlimited high-order scheme (the limiter keeps the run bounded): two oracle runs whose initial
fields differ by ONE ULP separate by orders of magnitude within a few steps.  The first step alone turns the ulp into
~1e-8 -- the conditioning of the order-6 Bernstein mass matrix -- and the separation then grows ~2x per step on average
(three orders of magnitude in ten steps; O(0.1) after 25).  At the step `dt / (2 p + 1)` the same pair stays together (slow,
linear drift), and so does p = 3 at the reference's step.
This is the CPU-side evidence for bench.py's two p = 6 `mass_check` legs: the growth the GPU
runs show is the scheme's, not the kernel's.
"""
import numpy as np
import pytest

from oracle.remhos_oracle import Config, Remhos


def _pair(p, rs, dt_scale, steps):
    cfl = Remhos(Config(mesh="periodic-cube", rs=rs, order=p, problem=10, dt=-1.0, t_final=0.5, lo=5)).dt
    # (local mass solve: the reference's own -pa rule, DGMassInverse's Jacobi-PCG stopped at abs 1e-8, remhos_ho.cpp:79-80, with
    # the product's completion -- the rule bench.py times; five times cheaper per step here than the dense exact solve, which
    # shows the same growth: 2.6e-8 -> 2.7e-5 in 11 steps)
    runs = [Remhos(Config(mesh="periodic-cube", rs=rs, order=p, problem=10, dt=cfl * dt_scale, t_final=0.5, lo=5, ho_solve="pa"))
            for _ in range(2)]
    runs[1].u = np.nextafter(runs[1].u, np.inf)  # one ulp
    sep = []
    for _ in range(steps):
        for r in runs:
            r.step(r.dt)
        sep.append(float(np.abs(runs[0].u - runs[1].u).max()))
    return sep


def test_p6_reference_step_amplifies_a_rounding_error():
    sep = _pair(6, 1, 1.0, 8)
    print("p = 6, dt = CFL rule:", " ".join(f"{s:.1e}" for s in sep))
    assert 1e-10 < sep[0] < 1e-6  # the mass solve's conditioning: one ulp -> ~1e-8 after a single step
    # measured: 2.1e-8 2.5e-8 4.5e-8 4.8e-8 1.7e-7 3.0e-7 5.2e-7 2.1e-6 9.8e-6 2.9e-5 (the exact solve: 0.29 after 25 steps)
    assert sep[-1] > 50.0 * sep[0]  # (eight steps: 2.1e-8 -> 2.1e-6)


def test_p6_stable_step_and_p3_do_not():
    sep = _pair(6, 1, 1.0 / 13.0, 6)
    print("p = 6, dt = CFL / (2 p + 1):", " ".join(f"{s:.1e}" for s in sep))
    assert max(sep) < 1e-7 and sep[-1] < 10.0 * sep[0]  # (the exact solve: 2.4e-9 ... 1e-8 after 14 steps: linear drift)
    sep3 = _pair(3, 1, 1.0, 5)
    print("p = 3, dt = CFL rule:", " ".join(f"{s:.1e}" for s in sep3))
    assert max(sep3) < 1e-11  # measured: 4e-13 and flat
