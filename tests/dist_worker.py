"""Worker of tests/test_dist_gloo.py: one rank of a box-partitioned run, gloo backend, kernels
through the g++ emulation build.  Writes its block of the final field to an .npz file."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    out_dir, mesh, rs, p, prob, steps, px, py, pz = sys.argv[1:10]
    rs, p, prob, steps, part = int(rs), int(p), int(prob), int(steps), (int(px), int(py), int(pz))
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == part[0] * part[1] * part[2]
    from remhos_amd.capi import load_library
    from remhos_amd.case import Case, bind_driver, make_config
    from remhos_amd.stepper import Stepper

    lib = bind_driver(load_library(os.path.join(ROOT, "tests", "emu", "librmh_emu.so")))
    # optional: LO solver, bounds type / dt control, dt and t_final (RMH_TEST_OPTS="lo,bt,dtc,dt,tf")
    lo, bt, dtc, dt, tf = 5, 0, 0, -1.0, 0.5
    if os.environ.get("RMH_TEST_OPTS"):
        a = os.environ["RMH_TEST_OPTS"].split(",")
        lo, bt, dtc, dt, tf = int(a[0]), int(a[1]), int(a[2]), float(a[3]), float(a[4])
    case = Case(lib, make_config(mesh, rs, p, prob, dt, tf, lo_type=lo, part=part, rank=rank, bounds_type=bt, dt_control=dtc))
    st = Stepper(lib, case, device="cpu", dist=dist, fused=True)
    nsteps = st.run(max_steps=steps)
    mass, umax = st.local_mass_and_max()
    red = torch.tensor([mass], dtype=torch.float64)
    dist.all_reduce(red, op=dist.ReduceOp.SUM)  # MPI_Allreduce of remhos.cpp:1412
    mx = torch.tensor([umax], dtype=torch.float64)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)  # remhos.cpp:1415
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gid=case.owned_gid, u=st.x.numpy(), mass=red.numpy(),
             umax=mx.numpy(), info=np.array([nsteps, st.repeats, st.dt]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
