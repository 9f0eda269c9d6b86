"""The shipped executable, end to end.  The reference's regression suites drive the *binary* and read its report
(autotest/test.sh:27-66 greps the printed masses; remhos_tests.cpp:109-180 runs `remhos` with the option strings of its
table): these tests do the same with remhos_amd/remhos_amd_run -- a child process on the command lines of README.md, its
stdout parsed for `Final mass u` / `Max value u` -- and check `-save`'s sltn_final.gf, written from a device-computed
field, against the CPU oracle's u."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "remhos_amd", "remhos_amd_run")
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))


def run_binary(args, cwd=None):
    assert os.path.exists(EXE), "remhos_amd/remhos_amd_run is not built (python -c 'import __graft_entry__ as g; g.build()')"
    p = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=600, cwd=cwd)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    return p.stdout


def printed(out, label):
    m = re.search(rf"^{re.escape(label)}\s*([-+0-9.eE]+)\s*$", out, re.M)
    assert m, (label, out)
    return float(m.group(1))


def ctest_entry(prefix):
    return next(e for e in KAT["ctest"] if e["name"].startswith(prefix))


def sig10(x):
    return float(f"{x:.10g}")


@pytest.mark.parametrize("extra", [[], ["-pa"]], ids=["exact", "pa"])
def test_binary_ctest7_command_line(extra):
    """README's first command line = the reference's ctest #7 (remhos_tests.cpp:81-86: cube01_hex -rs 3 -o 3, one step)."""
    e = ctest_entry("ctest7")
    out = run_binary(["-m", "data/cube01_hex.mesh", "-p", 10, "-rs", 3, "-o", 3, "-dt", -1, "-tf", 0.5, "-ms", 1,
                      "-ho", 3, "-lo", 5, "-fct", 2] + extra)
    assert printed(out, "Final mass u:") == sig10(e["mass"])
    assert int(re.search(r"time step: (\d+)", out).group(1)) == 1
    assert int(re.search(r"Number of unknowns: (\d+)", out).group(1)) == 16**3 * 4**3  # cube01_hex is 2 x 2 x 2: -rs 3 -> 16^3 hex x 4^3 dofs
    assert printed(out, "FOM wall (everything included):") > 0
    assert "FOM RHS:" in out and "FOM INV:" in out and "FOM LO:" in out and "FOM FCT:" in out  # remhos.cpp:1938-1952


def test_binary_ctest1_command_line_2d():
    """README's second command line = ctest #1 (inline-quad -rs 4 -o 3, 5 steps): dim = 2 kernels through the binary."""
    e = ctest_entry("ctest1")
    out = run_binary(["-m", "data/inline-quad.mesh", "-p", 14, "-rs", 4, "-o", 3, "-dt", -1, "-tf", 0.5, "-ms", 5,
                      "-ho", 3, "-lo", 5, "-fct", 2])
    assert printed(out, "Final mass u:") == sig10(e["mass"])
    assert int(re.search(r"time step: (\d+)", out).group(1)) == 5


def test_binary_autotest_line():
    """one line of autotest/test.sh through the binary: periodic-cube transport -ho 3 -lo 4 -fct 2 (out_baseline.dat:66-69)"""
    e = next(a for a in KAT["autotest"] if a["name"].startswith("periodic-cube transport -ho 3 -lo 4"))
    out = run_binary(["-m", "data/periodic-cube.mesh", "-p", 0, "-rs", 1, "-o", 2, "-dt", e["dt"], "-tf", e["t_final"],
                      "-ho", 3, "-lo", 4, "-fct", 2])
    assert printed(out, "Final mass u:") == e["mass"]
    assert printed(out, "Max value u:") == e["max"]


def test_binary_verify_bounds_passes():
    """-vb (remhos.cpp:324): the same run with the reference's in-loop guards on -- same printed numbers, exit code 0 -- through the
    one-kernel stage, the reference's call sequence (-lo 4 at a step inside its CFL limit), and product remap with an IDP solver."""
    lo4 = ["-m", "data/cube01_hex.mesh", "-p", 10, "-rs", 2, "-o", 3, "-dt", 0.01, "-tf", 0.5, "-ms", 3, "-ho", 3, "-lo", 4, "-fct", 2]
    ps = ["-m", "data/cube01_hex.mesh", "-p", 10, "-rs", 0, "-o", 2, "-dt", 0.02, "-tf", 0.5, "-ms", 3, "-ho", 3, "-lo", 5, "-fct", 2, "-ps", "-s", 13]
    for args in (lo4, lo4 + ["-unfused"], ps, ps + ["-unfused"]):
        a, b = run_binary(args), run_binary(args + ["-vb"])
        for label in ("Final mass u:", "Max value u:"):
            assert printed(a, label) == printed(b, label), (args, label)


@pytest.mark.parametrize("extra,lo,info", [(["-unfused"], 4, "LimitMult LO u"), ([], 4, "LimitMult FCT solution u"), (["-unfused"], 5, "LimitMult LO u")],
                         ids=["sequence-lo4", "one-kernel-lo4", "sequence-lo5"])
def test_binary_verify_bounds_aborts_like_the_reference(extra, lo, info):
    """-vb has to FAIL where the reference's would: on cube01_hex -rs 2 -o 3 the CFL step of -dt -1 is too long for the LO solvers on
    the second RK stage of the first step -- the CPU oracle's LO update leaves the dof bounds there (-lo 4: dof 1322 by 7.27e-05;
    -lo 5: dof 3204 by 2.82e-07), and with an LO update out of bounds ClipScale's rescaling step cannot hold them either.  The run
    aborts like MFEM_ABORT (remhos.cpp:1570, 1590) with the reference's message, naming the dof the ORACLE finds."""
    from oracle.remhos_oracle import Config, Remhos, check_violation

    r = Remhos(Config(mesh="cube01_hex", rs=2, order=3, problem=10, dt=-1.0, t_final=0.5, lo=lo, max_steps=1))
    keep = {}
    k = r.stage(r.u, 0.0, r.dt, keep)
    assert check_violation(r.u, keep["umin"], keep["umax"], dt=r.dt, du=keep["du_lo"])["count"] == 0  # (the first stage is clean)
    y = r.u + r.dt * k
    r.stage(y, r.dt, r.dt, keep)
    want = check_violation(y, keep["umin"], keep["umax"], dt=r.dt, du=keep["du_lo" if "LO" in info else "du"])
    assert want["count"] > 0 and want["over"] > 1e-7
    p = subprocess.run([EXE, "-m", "data/cube01_hex.mesh", "-p", "10", "-rs", "2", "-o", "3", "-dt", "-1", "-tf", "0.5", "-ms", "3", "-ho", "3",
                        "-lo", str(lo), "-fct", "2", "-vb"] + extra, capture_output=True, text=True, timeout=600)
    assert p.returncode == -6, (p.returncode, p.stdout[-1000:], p.stderr[-1000:])  # SIGABRT
    assert "Aborted due to bounds violation." in p.stderr
    m = re.search(rf"{info} bounds violation: (\d+) (\S+) (\S+) (\S+)", p.stdout)
    assert m, p.stdout[-1000:]
    assert int(m.group(1)) == want["first"]
    assert abs(float(m.group(3)) - want["first_value"]) < 1e-9 and abs(float(m.group(4)) - want["first_max"]) < 1e-12
    n = re.search(r"\((\d+) dofs out of bounds; largest overshoot (\S+),", p.stdout)
    assert int(n.group(1)) == want["count"] and abs(float(n.group(2)) - want["over"]) < 1e-3 * want["over"]


def test_binary_rejects_what_it_does_not_implement():
    p = subprocess.run([EXE, "-ho", "1"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "implements" in p.stderr
    p = subprocess.run([EXE, "-m", "data/star-q2.mesh", "-ms", "1"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "unknown lattice mesh" in p.stderr


def read_gf(path):
    tok = open(path).read().split()
    assert tok[:2] == ["FiniteElementSpace", "FiniteElementCollection:"]
    return tok[2], np.array(tok[7:], dtype=np.float64)


@pytest.mark.parametrize("mesh,rs,p,prob,steps", [("cube01_hex", 1, 3, 10, 3), ("periodic-cube", 1, 2, 0, 4)])
def test_binary_save_writes_the_device_field(tmp_path, mesh, rs, p, prob, steps):
    """`-save` (remhos.cpp:1365-1380): sltn_final.gf holds the field the DEVICE computed -- compared entry by entry with the
    oracle's u after the same steps (the file prints 16 significant digits); sltn_init.gf is the projected initial field."""
    from oracle.remhos_oracle import Config, Remhos
    from remhos_amd.capi import load_library
    from remhos_amd.case import Case, bind_driver, make_config

    dt, tf = (-1.0, 0.5) if prob >= 10 else (0.01, 0.5)
    out = run_binary(["-m", f"data/{mesh}.mesh", "-p", prob, "-rs", rs, "-o", p, "-dt", dt, "-tf", tf, "-ms", steps,
                      "-ho", 3, "-lo", 5, "-fct", 2, "-save"], cwd=tmp_path)
    for f in ("meshHO_init.mesh", "sltn_init.gf", "meshHO_final.mesh", "sltn_final.gf"):
        assert (tmp_path / f).exists(), f
    r = Remhos(Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=dt, t_final=tf, lo=5, max_steps=steps))
    u0 = r.u.copy()
    res = r.run()
    assert res["steps"] == steps
    case = Case(bind_driver(load_library()), make_config(mesh, rs, p, prob, dt, tf))
    order = np.argsort(case.owned_gid)  # the file is written in global element order
    fec, vals = read_gf(tmp_path / "sltn_final.gf")
    assert fec == f"L2_T2_3D_P{p}"
    got = vals.reshape(case.ne_owned, -1)
    # the oracle's element order is the case builder's (tests/test_gpu_kat.py compares them entry by entry)
    assert np.abs(got - r.u[order]).max() < 1e-10
    assert printed(out, "Final mass u:") == sig10(res["mass"])
    _, v0 = read_gf(tmp_path / "sltn_init.gf")
    assert np.abs(v0.reshape(case.ne_owned, -1) - u0[order]).max() < 1e-13
    # the final mesh is the initial one moved to the end time (remap) / unchanged (transport)
    m0, m1 = (tmp_path / "meshHO_init.mesh").read_text(), (tmp_path / "meshHO_final.mesh").read_text()
    assert (m0 != m1) == (prob >= 10)
