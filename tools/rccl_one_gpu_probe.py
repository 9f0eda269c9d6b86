"""Probe (validation aid): can the in-library RCCL exchange be exercised on a ONE-GPU box?  Two ranks, both on
cuda:0; the library creates its own communicator from a unique id passed through a gloo group.  RCCL normally
refuses two ranks on one device ("Duplicate GPU detected"); the probe reports what happened either way.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/rccl_one_gpu_probe.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remhos_amd.capi import load_library  # noqa: E402
from remhos_amd.case import Case, bind_driver, make_config  # noqa: E402
from remhos_amd.stepper import Stepper  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = bind_driver(load_library())
    case = Case(lib, make_config("periodic-cube", 2, 3, 10, -1.0, 0.5, part=(2, 1, 1), rank=rank))
    st = Stepper(lib, case, device="cuda:0")  # no dist: the transport is set up by hand below
    box = [st.ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    try:
        st.ctx.comm_init(box[0], world, rank)
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: rmh_comm_init failed: {e}", flush=True)
        dist.barrier()
        return
    st.transport = "rccl"
    for _ in range(3):
        st.step(st.dt)
    torch.cuda.synchronize()
    red = st.ctx.allreduce([float(st.x.sum())], "sum")
    blocks = [None] * world
    dist.all_gather_object(blocks, (case.owned_gid.copy(), st.x.cpu().numpy()))
    if rank == 0:
        gid = np.concatenate([b[0] for b in blocks])
        un = np.concatenate([b[1] for b in blocks])[np.argsort(gid)]
        c1 = Case(lib, make_config("periodic-cube", 2, 3, 10, -1.0, 0.5))
        s1 = Stepper(lib, c1, device="cuda:0")
        for _ in range(3):
            s1.step(s1.dt)
        torch.cuda.synchronize()
        u1 = s1.x.cpu().numpy()
        print("RCCL exchange inside the library, 2 ranks on one GPU: bitwise equal to the single-rank run:",
              bool(np.array_equal(un, u1)), "| rmh_allreduce sum", red[0], "vs", float(u1.sum()), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
