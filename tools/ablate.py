"""Diagnostic: HO kernel time vs local-solve tolerance (CG share) on the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper
lib = bind_driver(load_library(os.environ.get("RMH_LIB")))
rs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
st = Stepper(lib, Case(lib, make_config("periodic-cube", rs, 3, 10, -1.0, 0.5)), device="cuda:0")
for _ in range(2): st.step(st.dt)
def timeit(label, n=20):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): st.ctx.ho_apply(st.x, st.k)
    e1.record(); torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1)/n:8.4f} ms  cg iters {st.ctx.last_cg_iters()}")
st.ctx.setup(0.3)
timeit("full (rel 1e-14)")
st.ctx.set_mass_tol(1e-7); timeit("rel 1e-7")
st.ctx.set_mass_tol(1e-3); timeit("rel 1e-3")
st.ctx.set_mass_tol(1.0); timeit("rel 1 (no PCG iterations)")
st.ctx.set_mass_tol(1e-14, 0.0, 1); timeit("max_iter 1")
st.ctx.set_mass_tol(1e-14, 0.0, 2); timeit("max_iter 2")
st.ctx.set_mass_tol(1e-14, 0.0, 3); timeit("max_iter 3")
