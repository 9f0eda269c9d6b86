"""Development aid: run the first RK stage of a case over and over on the SAME input (valid data for as long as wanted)
so that clocks and power can be sampled with rocm-smi while the stage kernel is the only thing running.

    python tools/power_probe.py [--order 3 --rs 5 --lo 5 --seconds 12] name ...     (remhos_amd/librmh_<name>.so; main = librmh.so)
"""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remhos_amd.capi import load_library
from remhos_amd.case import Case, bind_driver, make_config
from remhos_amd.stepper import Stepper

ap = argparse.ArgumentParser()
ap.add_argument("--order", type=int, default=3)
ap.add_argument("--rs", type=int, default=5)
ap.add_argument("--lo", type=int, default=5)
ap.add_argument("--seconds", type=float, default=12.0)
ap.add_argument("names", nargs="+")
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception:  # noqa: BLE001
            break
        sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
        pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
        if sclk and pw:
            out.append((int(sclk.group(1)), float(pw.group(1))))
        time.sleep(0.5)


for name in args.names:
    path = os.path.join(root, "remhos_amd", f"librmh_{name}.so" if name not in ("", "main") else "librmh.so")
    lib = bind_driver(load_library(path))
    case = Case(lib, make_config("periodic-cube", args.rs, args.order, 10, -1.0, 0.5, lo_type=args.lo))
    st = Stepper(lib, case, device="cuda:0")
    st.step(case.dt)
    u = st.x.clone()
    out = torch.empty_like(u)
    st.ctx.setup(0.4)  # a deformed mesh (at t = 0 the Jacobi-preconditioned mass solve converges at once)
    for _ in range(5):
        st.ctx.stage_fused(u, case.dt, out, x_base=u, a=0.0, b=1.0, dt_rk=case.dt)
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    n = 0
    t0 = time.perf_counter()
    th.start()
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(50):
            st.ctx.stage_fused(u, case.dt, out, x_base=u, a=0.0, b=1.0, dt_rk=case.dt)
        torch.cuda.synchronize()
        n += 50
    el = time.perf_counter() - t0
    stop.set()
    th.join()
    busy = [s for s in samples[2:-1]] or samples
    nd = case.ne_global * case.ndof
    print(f"{name:10s} {1e3 * el / n:7.4f} ms/stage  {1e-6 * nd * n / el:9.1f} MDOFs*stage/s  "
          f"sclk {sum(s[0] for s in busy) / max(1, len(busy)):6.0f} MHz  power {sum(s[1] for s in busy) / max(1, len(busy)):6.0f} W  "
          f"cg {st.ctx.last_cg_iters()}  ({len(busy)} samples, finite {bool(torch.isfinite(out).all())})", flush=True)
    st.close()
    del st, case
    torch.cuda.empty_cache()
