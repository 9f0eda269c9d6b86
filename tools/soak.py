import sys, time
sys.path.insert(0, ".")
import torch
from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper
lib = bind_driver(load_library())
for lo in (5, 4):
    case = Case(lib, make_config("periodic-cube", 4, 3, 10, -1.0, 0.5, lo_type=lo))
    st = Stepper(lib, case, device="cuda:0")
    m0, _ = st.local_mass_and_max(0.0)
    t0 = time.time()
    n = st.run()  # the whole remap: pseudo-time 0 -> 1
    torch.cuda.synchronize()
    el = time.time() - t0
    m1, umax = st.local_mass_and_max()
    umin = float(st.x.min())
    print(f"lo {lo}: {n} steps in {el:.2f} s, {1e-6 * case.u0.size * 3 * n / el:.0f} MDOFs*stage/s, mass {m0:.15g} -> {m1:.15g} (loss {abs(m1-m0)/m0:.2e}), min {umin:.3e}, max {umax:.12f}, cg iters {st.ctx.last_cg_iters()}")
    assert umin > -1e-12 and umax < 1 + 1e-12
    st.close()
