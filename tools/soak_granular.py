"""Development aid (GPU): whole-remap soak of the GRANULAR paths (reference call sequence; HO kernel + fused limiter) against the
one-kernel stage -- same case, pseudo-time 0 -> 1: final masses, extrema and the field must agree to round-off growth."""
import sys
import time

sys.path.insert(0, ".")
import torch

from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper

lib = bind_driver(load_library())
CASES = ((3, 4, 5), (2, 4, 5), (4, 3, 5), (6, 3, 5), (5, 3, 5), (3, 3, 4), (6, 2, 4))
for order, rs, lo in (CASES[3:] if "--tail" in sys.argv else CASES):
    case = Case(lib, make_config("periodic-cube", rs, order, 10, -1.0, 0.5, lo_type=lo, pa=1))
    res = {}
    for name, kw in (("one-kernel", dict()), ("ho+limiter", dict(one_kernel=False)), ("call-sequence", dict(fused=False))):
        st = Stepper(lib, case, device="cuda:0", **kw)
        m0, _ = st.local_mass_and_max(0.0)
        t0 = time.time()
        if order >= 5:
            # the reference's p-independent CFL step is beyond the stability limit of the unlimited HO scheme at these orders:
            # runs a rounding error apart separate ~10x per step -- compare at the stable step dt / (2p + 1)
            st.dt = case.dt / (2 * order + 1)
            n = st.run(max_steps=100)
        else:
            n = st.run()
        torch.cuda.synchronize()
        el = time.time() - t0
        m1, umax = st.local_mass_and_max()
        res[name] = (st.x.clone(), m1)
        print(f"p {order} rs {rs} lo {lo} {name:14s}: {n} steps, {1e-6 * case.u0.size * 3 * n / el:7.0f} MDOFs*stage/s, mass loss {abs(m1 - m0) / m0:.2e}, "
              f"min {float(st.x.min()):.3e}, max {umax:.12f}", flush=True)
        st.close()
    ref, mref = res["one-kernel"]
    for name in ("ho+limiter", "call-sequence"):
        x, m = res[name]
        d = float((x - ref).abs().max())
        print(f"     {name:14s} vs one-kernel: max |dx| {d:.2e}, mass {abs(m - mref) / abs(mref):.2e}")
        assert abs(m - mref) <= 1e-12 * abs(mref) and d < (1e-7 if order >= 5 else 1e-8)
