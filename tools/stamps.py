"""Diagnostic (never part of the product build): per-phase cycle shares of ho_kernel2 from a
-DRMH_STAMPS build of the library (remhos_amd/librmh_stamps.so).

The deltas of thread 0 are accumulated in LDS and written once per workgroup at the end of the kernel (the first
version issued a global atomic per stamp, which the next s_waitcnt vmcnt(0) of the workgroup then waited for:
phantom waits worth a third of the kernel)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper
lib = bind_driver(load_library(os.path.join("remhos_amd", os.environ.get("RMH_STAMPS_LIB", "librmh_stamps.so"))))
rs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 100
order = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lo = int(sys.argv[4]) if len(sys.argv) > 4 else 5
nb = {1: 16, 2: 9, 3: 7, 4: 1, 5: 1, 6: 1}[order]
if lo != 5:
    nb = {2: 9, 3: 7, 4: 1, 5: 1, 6: 1}[order]
st = Stepper(lib, Case(lib, make_config("periodic-cube", rs, order, 10, -1.0, 0.5, lo_type=lo)), device="cuda:0")
if maxit > 0:
    st.ctx.set_mass_tol(1e-14, 0.0, maxit)  # converged (or capped) solve
else:
    st.ctx.set_mass_tol(0.0, 1e-8, 100)  # maxit = 0: the -pa rule (DGMassInverse's abs 1e-8 + completion)
    st.ctx.set_mass_completion(True, True)
for _ in range(2): st.step(st.dt)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.rmh_debug_stamps(buf, 1)
for _ in range(4): st.step(st.dt)
torch.cuda.synchronize()
lib.rmh_debug_stamps(buf, 0)
names = {10: "I: sA write+barrier+flag", 11: "I: x-leg", 12: "I: column", 13: "I: y-back", 14: "I: dot1+update", 15: "I: dot2", 0: "A loads", 1: "B T1+U1 pencils", 2: "B face rows", 3: "C column", 4: "F y-leg", 5: "G dof x-leg+faces", 6: "I PCG total(excl. split)", 7: "J back+stores", 8: "I: x-back", 9: "I: tail update/loop", 16: "J: back-transform", 21: "K: stencil -> LDS", 22: "K: mass dot", 17: "K: vol dot", 18: "K: bounds + clip", 19: "K: pos/neg dots", 20: "K: scale + stores", 7: "K: extrema + end", 24: "B: subcell pass", 25: "B: traces -> LDS + barrier", 26: "B: face rows (only)", 27: "G: x-leg + faces (only)", 28: "G: z GL -> Bernstein", 29: "G: RD element sums", 30: "G: RD extrema + chunk sums", 31: "C: column pass (before the combine of split columns)"}
tot = sum(buf[k] for k in range(32))
nblk = 12 * ((st.case.ne_owned + nb - 1) // nb)
for k in [k for k in range(32) if buf[k]]:
    print(f"{names.get(k, 'stamp ' + str(k)):28s} {buf[k]/nblk:10.0f} cycles/WG  {100.0*buf[k]/tot:5.1f}%")
print("total per WG", tot / nblk)
