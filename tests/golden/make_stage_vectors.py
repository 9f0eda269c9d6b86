#!/usr/bin/env python3
"""Generates tests/golden/stage_*.npz: inputs and expected outputs of ONE RK stage (HO -> LO -> bounds -> FCT) on
small meshes, computed by the CPU oracle (oracle/remhos_oracle.py, itself pinned by the reference's known answers in
tests/golden/reference_kat.json).  The reference holds no per-stage vectors ("parity unpinned at stage
granularity"), so these are the committed fixtures the GPU parity tests and the CPU port are checked against.

    python tests/golden/make_stage_vectors.py [name ...]     # rewrites the .npz files (deterministic: no RNG)

Stored per case: the C-ABI layout (x0, vel [ne][3][27], face_nbr [ne][6], stencil27 [ne][27], subcell_vel), the
stage input u, t, dt and the oracle's du_ho, du_lo, umin, umax, du, lumped mass m.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.remhos_oracle import Config, Remhos  # noqa: E402
from tests.helpers import layout_from_oracle, perturbed  # noqa: E402

CASES = [
    # name, mesh, rs, order, problem, lo, bounds_type, t
    ("remap_p2_lo5", "cube01_hex", 1, 2, 10, 5, 0, 0.3),
    ("remap_p3_lo5", "cube01_hex", 1, 3, 10, 5, 0, 0.5),
    ("transport_p3_lo4", "periodic-cube", 0, 3, 0, 4, 0, 0.0),
    ("remap_p2_lo4_bt1", "cube01_hex", 1, 2, 10, 4, 1, 0.4),
    ("remap_p3_lo3", "periodic-cube", 0, 3, 10, 3, 0, 0.6),
    ("remap_p4_lo5", "cube01_hex", 0, 4, 10, 5, 0, 0.3),
    # the -pa rule of the local mass solve: DGMassInverse's abs 1e-8 (remhos_ho.cpp:79-80) + the product's completion steps
    # (oracle: ho_solve = "pa"); the fixture carries mass_solve = "pa" and the oracle's PCG iteration count
    ("remap_p3_lo5_pa", "cube01_hex", 1, 3, 10, 5, 0, 0.5),
    # the high orders of BASELINE configs[2] (round 4)
    ("remap_p5_lo5", "cube01_hex", 0, 5, 10, 5, 0, 0.4),
    ("remap_p6_lo5", "periodic-cube", 0, 6, 10, 5, 0, 0.5),
    ("remap_p6_lo4", "cube01_hex", 0, 6, 10, 4, 0, 0.3),
    ("remap_p6_lo5_pa", "periodic-cube", 0, 6, 10, 5, 0, 0.5),
]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])  # (names given: write only those files -- the others stay byte for byte what is committed)
    for name, mesh, rs, p, prob, lo, bt, t in CASES:
        if only and name not in only:
            continue
        pa = name.endswith("_pa")
        cfg = Config(mesh=mesh, rs=rs, order=p, problem=prob, dt=0.01, t_final=0.7, lo=lo, bounds_type=bt, ho_solve="pa" if pa else "exact")
        r = Remhos(cfg)
        r.refine_steps = 2
        x0, vel, nbr, st = layout_from_oracle(r)
        u = perturbed(r.u)
        keep = {}
        du = r.stage(u, t, cfg.dt, keep)
        sub = np.zeros((0,))
        if lo == 4:
            sv = r.Vs if r.exec_mode == 1 else r.vel(r.Xs0)
            sub = np.ascontiguousarray(sv.transpose(0, 2, 1))
        np.savez_compressed(
            os.path.join(here, f"stage_{name}.npz"), order=p, exec_mode=r.exec_mode, lo=lo, bounds_type=bt, t=t, dt=cfg.dt,
            x0=x0, vel=vel, face_nbr=nbr, stencil27=st, subcell_vel=sub, u=u, du_ho=keep["du_ho"], du_lo=keep["du_lo"],
            umin=keep["umin"], umax=keep["umax"], du=du, m=keep["m"], mass_solve="pa" if pa else "exact",
            cg_iters=getattr(r, "cg_iters", 0) if pa else 0)
        print(name, u.shape, float(np.abs(du).max()))


if __name__ == "__main__":
    main()
