import sys, time
sys.path.insert(0, ".")
import torch
from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper
lib = bind_driver(load_library())
case = Case(lib, make_config("periodic-cube", 4, 3, 10, -1.0, 0.5))
st = Stepper(lib, case, device="cuda:0")
c, u, dt = st.ctx, st.x, st.dt
y = torch.empty_like(u)
ne = case.ne_owned
nh = 48**3 - 46**3
c.setup(0.3)
def run(split, n=60):
    for _ in range(5):
        c.stage_fused(u, dt, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if split:
            c.stage_fused_range(u, dt, y, nh, ne, False)
            c.stage_fused_range(u, dt, y, 0, nh, True)
        else:
            c.stage_fused(u, dt, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("one launch  %.4f ms" % run(False)); print("two launches %.4f ms (interior %d + halo %d elements)" % (run(True), ne - nh, nh)); print("one launch  %.4f ms" % run(False))
