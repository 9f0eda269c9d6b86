"""One-rank RCCL self-loop of the stage exchange at bench scale (rmhd_config.self_wrap): the grouped ncclSend /
ncclRecv of rmh_exchange_begin run on the GPU at hand.  Under `rocprofv3 --kernel-trace --stats` the RCCL device kernel
shows up beside the pack and stage kernels (summary kept under profiles/).

    python tools/rccl_selfloop.py [--rs 4 --order 3 --steps 10 --full-records]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper

ap = argparse.ArgumentParser()
ap.add_argument("--rs", type=int, default=4)
ap.add_argument("--order", type=int, default=3)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--full-records", action="store_true")
ap.add_argument("--comm-cus", type=int, default=0, help="run the stage kernels on a stream that leaves this many CUs to the exchange stream (rmh_stream_create_reserving)")
ap.add_argument("--no-overlap", action="store_true", help="exchange first, then ONE launch over all elements: the RCCL kernel alone on the GPU")
args = ap.parse_args()
if args.full_records:
    os.environ["RMH_COMPACT"] = "0"
lib = bind_driver(load_library())
out = {}
masked = None
if args.comm_cus > 0:
    import ctypes as C

    h = C.c_void_p()
    assert lib.rmh_stream_create_reserving(0, args.comm_cus, C.byref(h)) == 0, lib.rmh_last_error()
    masked = torch.cuda.ExternalStream(h.value, device="cuda:0")
    torch.cuda.set_stream(masked)  # (the Stepper hands torch's current stream to its context)
for wrap in (0, 1):
    case = Case(lib, make_config("periodic-cube", args.rs, args.order, 10, -1.0, 0.5, pa=1, self_wrap=wrap))
    st = Stepper(lib, case, device="cuda:0", overlap=not args.no_overlap)
    for _ in range(2):
        st.step(case.dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.step(case.dt)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    info = ""
    if wrap:
        _, sn, _, gn = st.ctx.exchange_buffers()
        info = f", transport {st.transport}, {'compact' if st.compact else 'full'} records, {8 * sn} B sent and {8 * gn} B received per stage, halo shell {case.ne_halo} of {case.ne_owned} elements"
    print(f"self_wrap {wrap} comm_cus {args.comm_cus}: {1e-6 * case.u0.size * 3 * args.steps / el:.0f} MDOFs*stage/s{info}", flush=True)
    out[wrap] = st.x.cpu().numpy()[np.argsort(case.owned_gid)]
    st.close()
print("bit-identical to the plain periodic run:", bool(np.array_equal(out[0], out[1])))
