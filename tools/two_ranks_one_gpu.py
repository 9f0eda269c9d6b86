"""Validation aid (not a benchmark): N ranks of the box-partitioned stepper on ONE GPU with the gloo backend, to
exercise Stepper's multi-rank code (ghost records, interior / halo ranges, dt control all-reduce) on real kernels
where only one device is available.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 tools/two_ranks_one_gpu.py
Every rank prints a checksum; rank 0 gathers the blocks and compares them bit for bit with a single-rank run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remhos_amd.capi import load_library  # noqa: E402
from remhos_amd.case import Case, bind_driver, make_config  # noqa: E402
from remhos_amd.stepper import Stepper  # noqa: E402

PART = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = bind_driver(load_library())
    lo = int(os.environ.get("LO", "5"))
    kw = dict(lo_type=lo)
    case = Case(lib, make_config("periodic-cube", 2, 3, 10, -1.0, 0.5, part=PART[world], rank=rank, **kw))
    st = Stepper(lib, case, device="cuda:0", dist=dist)
    for _ in range(3):
        st.step(st.dt)
    torch.cuda.synchronize()
    u = st.x.cpu().numpy()
    blocks = [None] * world
    dist.all_gather_object(blocks, (case.owned_gid.copy(), u))
    if rank == 0:
        gid = np.concatenate([b[0] for b in blocks])
        un = np.concatenate([b[1] for b in blocks])[np.argsort(gid)]
        c1 = Case(lib, make_config("periodic-cube", 2, 3, 10, -1.0, 0.5, **kw))
        s1 = Stepper(lib, c1, device="cuda:0")
        for _ in range(3):
            s1.step(s1.dt)
        torch.cuda.synchronize()
        u1 = s1.x.cpu().numpy()
        print("ranks", world, "lo", lo, "bitwise equal to the single-rank run:", bool(np.array_equal(un, u1)), "max diff", float(np.abs(un - u1).max()))
        assert np.array_equal(un, u1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
