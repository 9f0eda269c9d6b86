#!/usr/bin/env python3
"""Benchmark of the Remhos RK stage hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one RK3-SSP time step = 3 RK stages, each stage = HO (PA convection + upwind faces +
local mass solve) -> LO (mass-based average) -> overlap bounds -> ClipScale FCT -> RK update on
the whole mesh.  Metric (BASELINE.json): MDOFs x RK-stage / s = 1e-6 * global dofs * 3K / T,
T = max over ranks of the wall time of K steps (everything included: halo exchange, mass solve,
bounds, RK update), inputs resident in HBM.

Workload (config.workload): BASELINE.json configs[1] -- 3D periodic-cube remap (problem 10,
Taylor-Green mesh motion, erfc bump), p = 3, -pa -ho 3 -lo 5 -fct 2 (the combination the reference
itself allows on a device, remhos.cpp:391-397), refined to --rs levels (default 5: 884 736 hex,
56.6 M dofs).  For N > 1 the SAME global mesh is box-partitioned over the ranks (strong scaling);
each stage does one RCCL neighbour exchange of ghost-element values.

The same JSON line carries `roofline` for the dominant kernel (ho_kernel2<p, 1>, duration from HIP events
on the kernel's own stream inside the timed region) and `cpu_baseline` (the C++/OpenMP CPU port of
oracle/ timed on the host cores on a bounded sample of the same workload; rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6  # vendor FP64 vector peak of MI355X (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz); tools/ubench_fp64.hip sustains 60.5
PART = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}


def ho_alg_bytes_per_element(p):
    """Algorithmic HBM bytes of the HO kernel per element (phase 1,
    matrix-free variant): read u D^3, neighbour traces 6 D^2, Q2 nodes of x0 and v 2*81;
    write du_HO D^3."""
    D = p + 1
    return 8 * (2 * D**3 + 6 * D**2 + 162)


def stage_alg_bytes_per_dof(p):
    """SURVEY.md 8(d) matrix-free model for the whole stage: 8*(8 D^3 + 6 D^2 + 162 + 6 p^3)/D^3."""
    D = p + 1
    return 8.0 * (8 * D**3 + 6 * D**2 + 162 + 6 * p**3) / D**3


def cpu_baseline(lib, order, budget_s=15.0):
    """Time the CPU port (oracle/cpu_port.cpp: C++/OpenMP restatement of the reference's CPU
    partial-assembly stage, validated against the reference's known answers) on the host cores, on a
    bounded sample of the same workload: periodic-cube remap, same order, -rs 4 (110 592 hex) at p <= 3."""
    from oracle.cpu_port import CpuPort
    from remhos_amd.case import Case, make_config

    rs = 4 if order <= 3 else (3 if order <= 4 else 2)  # enough elements per host thread, bounded run time
    case = Case(lib, make_config("periodic-cube", rs, order, 10, -1.0, 0.5))
    cp = CpuPort(order, case.exec_mode, case.x0, case.vel, case.face_nbr, case.stencil27, case.u0)
    cp.step(case.dt)  # warm-up (first touch, thread pool start)
    stages, t0 = 0, time.perf_counter()
    while stages == 0 or (time.perf_counter() - t0 < budget_s and stages < 30):
        cp.step(case.dt)
        stages += 3
    el = time.perf_counter() - t0
    ndofs = case.u0.size
    return {
        "value": 1e-6 * ndofs * stages / el,
        "unit": "MDOFs*RK-stage/s",
        "cores": cp.threads,
        "kind": "port",
        "sample": f"oracle/cpu_port.cpp (C++/OpenMP, {cp.threads} threads): periodic-cube -rs {rs} -o {order} -p 10 "
                  f"-lo 5 -fct 2, {ndofs} dofs, {stages} RK stages in {el:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rs", type=int, default=5)
    ap.add_argument("--order", type=int, default=3)
    ap.add_argument("--mesh", default="periodic-cube")
    ap.add_argument("--problem", type=int, default=10)
    ap.add_argument("--lo", type=int, default=5, help="LO solver: 5 mass-based average (default), 4 subcell residual distribution")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="reference call sequence instead of the fused limiter")
    ap.add_argument("--ref-mass-tol", action="store_true",
                    help="local mass solve with the reference's DGMassInverse tolerances (abs 1e-8, rel 0: remhos_ho.cpp:79-80) "
                         "instead of rel 1e-14 (see DESIGN.md 4)")
    ap.add_argument("--two-kernels", action="store_true", help="HO kernel + fused limiter kernel instead of the one-kernel stage")
    args = ap.parse_args()

    import torch

    from remhos_amd.capi import load_library
    from remhos_amd.case import Case, bind_driver, make_config
    from remhos_amd.stepper import Stepper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if args.gpus not in PART:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the remhos_amd hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device(dev))

    lib = bind_driver(load_library())
    cfg = make_config(args.mesh, args.rs, args.order, args.problem, -1.0, 0.5, lo_type=args.lo, part=PART[args.gpus], rank=rank)
    t0 = time.perf_counter()
    case = Case(lib, cfg)
    st = Stepper(lib, case, device=dev, dist=dist, fused=not args.unfused, one_kernel=not args.two_kernels,
                 overlap=os.environ.get("RMH_NO_OVERLAP", "0") != "1")  # (escape hatch: exchange, then one launch)
    if args.ref_mass_tol:
        st.ctx.set_mass_tol(0.0, 1e-8, 100)
    setup_s = time.perf_counter() - t0
    global_dofs = case.ne_global * case.ndof
    dt = case.dt

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        st.step(dt)
    st.ctx.last_cg_iters()  # reset
    st.ctx.enable_timers(True)
    st.ctx.reset_timers()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.step(dt)
    barrier()
    elapsed = time.perf_counter() - t0
    tim = st.ctx.timers()
    st.ctx.enable_timers(False)
    cg_iters = st.ctx.last_cg_iters()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])

    stages = 3 * args.steps
    value = 1e-6 * global_dofs * stages / elapsed
    # sanity of the state after the run: mass conservation and bounds (not timed)
    mass, umax = st.local_mass_and_max()
    if dist is not None:
        red = torch.tensor([mass], dtype=torch.float64, device=dev)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        mass = float(red[0])

    if rank == 0:
        ho_avg_s = tim[0] / stages
        if st.one_kernel:
            # the dominant kernel is the whole stage: SURVEY 8(d) matrix-free per-dof figure
            ho_bytes = int(stage_alg_bytes_per_dof(args.order) * case.ne_owned * case.ndof)
            kname = f"rmh::ho_kernel2<{args.order}, {3 if args.lo in (3, 4) else 1}> (whole RK stage)"
        else:
            ho_bytes = ho_alg_bytes_per_element(args.order) * case.ne_owned
            kname = f"rmh::ho_kernel2<{args.order}, 0>" if os.environ.get("RMH_HO_KERNEL", "2") != "1" else f"rmh::ho_kernel<{args.order}, 0>"
        achieved = ho_bytes / ho_avg_s / 1e9
        traffic, fp64 = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_ho_kernel.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.mesh}-rs{args.rs}-o{args.order}-n{args.gpus}" + ("-stage" if st.one_kernel else "")
                ent = tj.get(key, {}) if args.lo == 5 else {}
                traffic = ent.get("hbm_bytes_per_launch")
                wi = ent.get("fp64_wave_insts_per_launch")
                if wi:
                    # the kernel's real bound: FP64 VALU.  Issued lane-flops = wave64 instructions x 64 lanes x (2 for FMA)
                    flops = 64.0 * (2.0 * wi["fma"] + wi["mul"] + wi["add"])
                    fp64 = {"bound": "fp64 valu", "flops_per_launch": flops, "achieved": flops / ho_avg_s / 1e12,
                            "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / ho_avg_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                            "fp64_share_of_valu_insts": (wi["fma"] + wi["mul"] + wi["add"]) / wi["all_valu"]}
            except Exception:
                traffic, fp64 = None, None
        out = {
            "metric": "MDOFs*RK-stage/s, 3D hex remap",
            "value": value,
            "unit": "MDOFs*RK-stage/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.mesh} -rs {args.rs} -o {args.order} -p {args.problem} {'remap' if args.problem >= 10 else 'transport'}, "
                            f"-pa -ho 3 -lo {args.lo} -fct 2, RK3-SSP"
                            + (" (BASELINE configs[1])" if (args.mesh, args.order, args.problem, args.lo) == ("periodic-cube", 3, 10, 5) else "")
                            + f"; {case.ne_global} hex, {global_dofs} dofs",
                "global_dofs": global_dofs,
                "elements": case.ne_global,
                "partition": "x".join(str(k) for k in PART[args.gpus]),
                "limiter": "reference call sequence" if args.unfused else ("inside the stage kernel" if st.one_kernel else "fused (LO avg + bounds + ClipScale + RK update)"),
                "dt": dt,
                "mass_cg_max_iters": cg_iters,
                "mass_tol": "abs 1e-8 (reference)" if args.ref_mass_tol else "rel 1e-14",
                "final_mass": mass,
                "max_value": umax,
                "setup_s": setup_s,
            },
            "roofline": {
                "kernel": kname,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "avg_launch_ms": 1e3 * ho_avg_s,
                "alg_bytes_per_launch": ho_bytes,
                "note": "FP64 VALU / latency bound at p=3 in matrix-free form (geometry recomputed per stage), not HBM bound; see DESIGN.md 3.1",
            },
            "roofline_fp64": fp64,
            "buckets_s": {"ho_rhs_plus_inv_or_stage": tim[0], "lo": tim[2], "fct_or_fused_limiter": tim[3]},
            # the reference's figures of merit (remhos.cpp:1918-1966): 1e-6 * dofs * stages / bucket time of rank 0; its
            # printed total uses T_rhs + T_LO + T_FCT.  Here RHS and INV are one kernel (bucket 0); with the one-kernel
            # stage everything is in bucket 0.  "value" above is the all-inclusive wall-clock figure.
            "fom_reference_style": {
                k: (1e-6 * global_dofs * stages / v if v > 0 else None)
                for k, v in (("rhs_plus_inv", tim[0]), ("lo", tim[2]), ("fct", tim[3]), ("total_rhs_lo_fct", tim[0] + tim[2] + tim[3]))
            },
            "stage_roofline": {
                "alg_bytes_per_dof": stage_alg_bytes_per_dof(args.order),
                "achieved_GBs": value * 1e6 * stage_alg_bytes_per_dof(args.order) / 1e9,
                "frac_of_hbm_peak": value * 1e6 * stage_alg_bytes_per_dof(args.order) / 1e9 / (HBM_PEAK_GBS * args.gpus),
            },
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lib, args.order)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    st.close()


if __name__ == "__main__":
    main()
