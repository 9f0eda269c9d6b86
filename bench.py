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
56.6 M dofs).  At N = 1 the same JSON line carries a "p6" block: BASELINE.json configs[2]
(periodic-cube -rs 4 -o 6, 37.9 M dofs), timed after the p = 3 region with the same K and W, a "transport" block (the
same mesh with -p 0: the periodic behaviour the reference pins), an "lo4" block (the p = 3 and p = 6 workloads with
the subcell residual-distribution LO solver, -lo 4, inside the stage kernel) and a "sustained" block (the p = 3 workload
for >= 200 steps with clock / power samples and the first-step / steady split).

Local mass solve (`--mass-solve`): "pa" (default) = what -pa means in the reference, DGMassInverse stopped at abs 1e-8
(remhos_ho.cpp:79-80), completed by one Jacobi step and the constant mode (rmh_set_mass_completion: no further mass
apply, every stage conserves the mass to round-off); "exact" = converged to rel 1e-14 (the reference's full-assembly
element inverse); "reference" = the literal abs 1e-8 rule without completion.  `mass_check` compares the timed run's
final mass and field with an untimed converged run of the same steps, and shows the drift of a round-off twin
(converged + constant mode) as the floor of that comparison.

N > 1 (one process per GPU, RCCL): `--scaling weak` (default) keeps one -rs 5 block per GPU -- the lattice is
refined once more in x (N = 2), x and y (N = 4), all three directions (N = 8: exactly -rs 6) -- and
`--scaling strong` box-partitions the SAME -rs 5 mesh.  Each stage does one neighbour exchange of ghost-element
records.  Started without WORLD_SIZE, `--gpus N` launches its N ranks itself (torch.distributed.run as a child
process, before anything in this process touches the GPU).

The LAST stdout line is the record: one compact JSON line (< 6 KB, numbers and short labels only, `compact_line`); the full result with every
explanatory string goes to bench_detail.json (DESIGN.md 5 explains each field).  The line carries `roofline` for the dominant kernel (ho_kernel2<p, 1>, duration from HIP events
on the kernel's own stream inside the timed region) and `cpu_baseline` (the C++/OpenMP CPU port of
oracle/ timed on the host cores on a bounded sample of the same workload; rank 0 at N = 1 only).
"""
import argparse
import copy
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
HBM_SUSTAINED_GBS = 6300.0  # what a plain copy sustains on this part (same guide; this library's streaming kernels reach 0.74-0.78 of 8 TB/s)
FP64_VALU_PEAK_TFLOPS = 78.6  # vendor FP64 vector peak of MI355X (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz); tools/ubench_fp64.hip sustains 60.5
PART = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}
KERNEL_SOURCES = ("rmh_ho2.hpp", "rmh_kernels.hpp", "rmh_tables.hpp")


def ho_alg_bytes_per_element(p):
    """Algorithmic HBM bytes of the HO kernel per element (phase 1,
    matrix-free variant): read u D^3, neighbour traces 6 D^2, Q2 nodes of x0 and v 2*81;
    write du_HO D^3."""
    D = p + 1
    return 8 * (2 * D**3 + 6 * D**2 + 162)


def stage_alg_bytes_per_dof(p, lo=5):
    """SURVEY.md 8(d) matrix-free model for the whole stage: 8*(8 D^3 + 6 D^2 + 162 + 6 p^3)/D^3.
    lo 4 (subcell residual distribution) adds the arrays that solver cannot derive from anything else on the path
    (SURVEY App. D): the sub-mesh velocities at the 3 p^3 subcell midpoints and at the 3 D^3 sub-mesh nodes (the
    instantaneous velocity field sampled there with the boundary zeroed, remhos.cpp:837-853 -- not the Q2 interpolant).
    NOT counted although the kernel reads them: the sub-mesh start positions (3 D^3: the Q2 map of the 27 nodes at i/p) and
    the face speed table (3 Q * Q doubles per FACE -- about 3 blocks per element on a periodic lattice -- a stored intermediate that replaces recomputation) -- they show
    in `traffic`, not in the model."""
    D = p + 1
    extra = (3 * p**3 + 3 * D**3) if lo == 4 else 0
    return 8.0 * (8 * D**3 + 6 * D**2 + 162 + 6 * p**3 + extra) / D**3


def kernel_source_hash():
    """sha256 over the kernel sources: stored PMC counters are only valid for the kernel they were measured on."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "remhos_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


MASS_SOLVE = {  # name -> (rel_tol, abs_tol, max_iter, jacobi_step, constant_mode), label
    "pa": ((0.0, 1e-8, 100, 1, 1), "abs 1e-8 (DGMassInverse, remhos_ho.cpp:79-80) + Jacobi step + constant mode"),
    "exact": ((1e-14, 0.0, 100, 0, 0), "rel 1e-14"),
    "reference": ((0.0, 1e-8, 100, 0, 0), "abs 1e-8 (reference)"),
    "roundoff-twin": ((1e-14, 0.0, 100, 0, 1), "rel 1e-14 + constant mode"),
}


COMPACT_LIMIT = 6144  # the driver keeps an ~8 KB tail of stdout: the final line must fit with room to spare


def _num(x, sig=7):
    """numbers of the compact line: `sig` significant digits (the full precision is in bench_detail.json)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):  # (masses and maxima are compared digit by digit with the reference's prints: keep 13)
        return {k: _num(v, 13 if k in ("final_mass", "max_value") else sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    return x


def _short_workload(w):
    """'periodic-cube -rs 5 -o 3 -p 10 remap, -pa ... (BASELINE configs[1]); 96x96x96 = ...' -> the part before ';'"""
    return w.split(";")[0].replace(", RK3-SSP", " RK3").replace("-pa -ho 3 ", "-pa ").strip() if isinstance(w, str) else w


def _roof(r):
    """the contract's roofline object, numbers only (what each field means: DESIGN.md 5)"""
    if not r:
        return None
    # (results of rounds <= 5 carried the contract's model under "model" and what binds under "bound")
    bound, binds = (r["model"], r["bound"]) if r.get("model") else (r["bound"], r.get("binds"))
    return {"kernel": r["kernel"].split(" (")[0], "bound": bound, "binds": binds, "achieved": r["achieved"], "peak": r["peak"],
            "unit": r["unit"], "frac": r["frac"], "traffic": r.get("traffic"), "dram_frac": r.get("dram_frac"), "of_bound": r.get("of_bound"),
            "avg_launch_ms": r["avg_launch_ms"], "alg_bytes_per_launch": r["alg_bytes_per_launch"]}


def _fp64(r):
    return {"achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
            "fp64_share_of_valu_insts": r["fp64_share_of_valu_insts"]} if r else None


def _mc(m):
    """mass_check -> numbers and verdicts only"""
    if not m:
        return None
    o = {"mass_rel_dev": m["mass_rel_dev"], "field_max_dev": m["field_max_dev"], "pass": m["pass"]}
    if m.get("chaotic_dt"):
        o["chaotic_dt"] = True
    if m.get("stable_dt"):
        s = m["stable_dt"]
        o["stable_dt"] = {"steps": s["steps"], "mass_rel_dev": s["mass_rel_dev"], "field_max_dev": s["field_max_dev"], "pass": s["pass"],
                          "value": s.get("value_at_this_dt"), "cg_iters": s.get("cg_iters")}
    return o


def _sub(b):
    """a secondary block of the line: value, ms_per_step, avg_launch_ms, frac, pass"""
    if not b:
        return None
    o = {"value": b["value"], "ms_per_step": b["ms_per_step"]}
    r = b.get("roofline") or {}
    o["avg_launch_ms"] = b.get("avg_launch_ms", r.get("avg_launch_ms"))
    o["frac"] = r.get("frac", b.get("roofline_hbm_model_frac"))
    if b.get("roofline_fp64"):
        o["fp64_frac"] = b["roofline_fp64"]["frac"]
    if r.get("traffic") is not None:
        o["traffic"] = r["traffic"]
    c = b.get("config") or {}
    if "workload" in c or "workload" in b:
        o["workload"] = _short_workload(c.get("workload", b.get("workload")))
    fm = c.get("final_mass", b.get("final_mass"))
    if fm is not None:
        o["final_mass"] = fm
    it = c.get("mass_cg_max_iters", b.get("mass_cg_max_iters"))
    if it is not None:
        o["cg_iters"] = it
    if b.get("mass_check"):
        o["mass_check"] = _mc(b["mass_check"])
        o["pass"] = b["mass_check"]["pass"]
    return o


def compact_line(out, detail_file="bench_detail.json"):
    """The ONE line the driver parses: the contract's fields + numbers-only sub-blocks, < COMPACT_LIMIT bytes.  `out` is the
    full result dict (written to `detail_file`); every explanatory string lives in DESIGN.md 5."""
    c = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": _short_workload(c.get("workload")), "global_dofs": c.get("global_dofs"), "elements": c.get("elements"),
                      "partition": c.get("partition"), "tile_rows": c.get("tile_rows"), "mass_solve": c.get("mass_solve"), "cg_iters": c.get("mass_cg_max_iters"),
                      "dt": c.get("dt"), "final_mass": c.get("final_mass"), "max_value": c.get("max_value")}
    if c.get("stage_loop"):
        line["config"]["stage_loop"] = c["stage_loop"]
    line["roofline"] = _roof(out.get("roofline"))
    line["roofline_fp64"] = _fp64(out.get("roofline_fp64"))
    if "rccl_ranks" in out:
        line["rccl_ranks"] = out["rccl_ranks"]
    if out.get("exchange"):
        x = out["exchange"]
        line["exchange"] = {"transport": str(x.get("transport", "")).split(" ")[0], "neighbour_ranks": x.get("neighbour_ranks"),
                            "send_bytes_per_stage_rank0": x.get("send_bytes_per_stage_rank0")}
    if out.get("strong"):
        line["strong"] = out["strong"]
    if out.get("mass_check"):
        line["mass_check"] = _mc(out["mass_check"])
    for k in ("p6", "transport", "cube01_p4"):
        if out.get(k):
            line[k] = _sub(out[k])
    if out.get("sustained"):
        s = out["sustained"]
        line["sustained"] = {k: s[k] for k in ("steps", "value", "ms_per_step", "first_step_ms", "median_step_ms", "p95_step_ms")}
        smi = s.get("smi") or {}
        if smi.get("sclk_MHz"):
            line["sustained"]["sclk_MHz_min"] = min(smi["sclk_MHz"])
            line["sustained"]["power_W_max"] = max(smi["power_W"])
    if out.get("lo4"):
        line["lo4"] = {k: dict(_sub(v) or {}, form=v.get("form")) for k, v in out["lo4"].items() if isinstance(v, dict)}
    if out.get("granular"):
        g = out["granular"]
        line["granular"] = {"reference_call_sequence": {k: g["reference_call_sequence"][k] for k in ("value", "ms_per_step")},
                            "ho_plus_fused_limiter": {k: g["ho_plus_fused_limiter"][k] for k in ("value", "ms_per_step")},
                            "ho_kernel_ms": g["ho_kernel"]["avg_launch_ms"],
                            "streaming": {k.replace("_kernel", ""): {"ms": v["avg_launch_ms"], "frac": v["frac"]}
                                          for k, v in g["streaming_kernels"].items()}}
    if out.get("cpp_loop"):
        line["cpp_loop"] = {k: out["cpp_loop"].get(k) for k in ("value", "ms_per_step", "avg_launch_ms", "vs_python_loop", "pass")}
    if out.get("configs0_2d"):
        line["configs0_2d"] = {k: {kk: v.get(kk) for kk in ("final_mass", "max_value", "value", "pass", "error") if kk in v}
                               for k, v in out["configs0_2d"].items() if isinstance(v, dict)}
    if out.get("cpu_baseline"):
        b = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": b["value"], "unit": b["unit"], "cores": b["cores"], "kind": b["kind"],
                                "sample": b.get("sample_short", b.get("sample", ""))[:160]}
    line["detail"] = detail_file
    line = _num(line)
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) >= COMPACT_LIMIT:  # never let the record outgrow the driver's tail again: drop the optional blocks, last first
        for k in ("configs0_2d", "granular", "sustained", "cpp_loop", "lo4", "cube01_p4", "transport", "mass_check", "exchange"):
            line.pop(k, None)
            txt = json.dumps(line, separators=(",", ":"))
            if len(txt) < COMPACT_LIMIT:
                break
    if len(txt) >= COMPACT_LIMIT:  # still too long (a long workload / sample string, many sub-blocks): the contract's fields only
        print(f"bench.py: compact record {len(txt)} B >= {COMPACT_LIMIT} B after dropping the optional blocks -- cutting it to the contract's "
              f"fields (everything is in {detail_file})", file=sys.stderr, flush=True)
        for k in ("p6", "strong", "roofline_fp64", "rccl_ranks"):
            line.pop(k, None)
        if isinstance(line.get("cpu_baseline"), dict):
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample", ""))[:80]
        if isinstance(line.get("config"), dict):
            line["config"] = {k: (v[:120] if isinstance(v, str) else v) for k, v in line["config"].items()}
        txt = json.dumps(line, separators=(",", ":"))
        assert len(txt) < COMPACT_LIMIT, len(txt)
    return txt


def emit(out, n_gpus):
    """Write the full result to bench_detail[_nN].json (repo root, and gpurun_out/ when it exists) and print the compact line
    as the LAST line of stdout."""
    name = "bench_detail.json" if n_gpus == 1 else f"bench_detail_n{n_gpus}.json"
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError as e:
                print(f"bench.py: could not write {os.path.join(d, name)}: {e}", file=sys.stderr)
    print(compact_line(out, name), flush=True)


_X2 = {}  # key -> measured HBM bytes per launch with the guide's FETCH_SIZE correction (x 2) instead of the calibrated one


def stored_counters(key, mass_tol, lo):
    """HBM traffic and FP64 instruction counts per launch from the rocprofv3 --pmc passes committed under profiles/
    (tools/profile.sh, tools/pmc_insts.sh + tools/update_traffic.py).  They cannot be collected inside this run (the
    counters need the profiler); an entry is used only if it was measured on the same kernel sources, mass tolerance
    and LO solver, otherwise the fields are null with the reason."""
    tpath = os.path.join(ROOT, "profiles", "traffic_ho_kernel.json")
    try:
        ent = json.load(open(tpath)).get(key)
    except Exception as e:  # noqa: BLE001
        return None, None, f"profiles/traffic_ho_kernel.json unreadable: {e}"
    if not ent:
        return None, None, f"no PMC entry '{key}' under profiles/"
    if ent.get("kernel_src_sha") != kernel_source_hash():
        return None, None, f"PMC entry '{key}' was measured on other kernel sources ({ent.get('kernel_src_sha')})"
    if ent.get("mass_tol", "rel 1e-14") != mass_tol or ent.get("lo", 5) != lo:
        return None, None, f"PMC entry '{key}' was measured with other solver options"
    _X2[key] = ent.get("hbm_bytes_per_launch_x2")
    return ent.get("hbm_bytes_per_launch"), ent.get("fp64_wave_insts_per_launch"), ent.get("source", "profiles/")


def cpu_baseline(lib, order, rs, mass_solve="pa", budget_s=12.0):
    """Time the CPU port (oracle/cpu_port.cpp: C++/OpenMP restatement of the reference's CPU partial-assembly stage,
    contractions specialised on the order, the HO loop vectorised across batches of 8 (AVX-512) or 4 (AVX2) elements,
    validated against the reference's known answers) on the host cores, one pinned thread per core, on a bounded sample
    of the same workload with the same local mass solve as the GPU run: the same mesh as the GPU when one RK step fits
    the time budget, one refinement level less otherwise.  Reports the reference-style bucket FOMs beside the
    wall-clock figure."""
    from oracle.cpu_port import CpuPort
    from remhos_amd.case import Case, make_config

    def run(rs_cpu, max_stages):
        case = Case(lib, make_config("periodic-cube", rs_cpu, order, 10, -1.0, 0.5))
        (rel, ab, _, jac, fix), _ = MASS_SOLVE[mass_solve]
        cp = CpuPort(order, case.exec_mode, case.x0, case.vel, case.face_nbr, case.stencil27, case.u0, rel_tol=rel, abs_tol=ab,
                     completion=bool(jac and fix))
        t0 = time.perf_counter()
        cp.step(case.dt)  # warm-up (first touch, thread pool start)
        warm = time.perf_counter() - t0
        cp.buckets(reset=True)
        stages, t0 = 0, time.perf_counter()
        while stages == 0 or (time.perf_counter() - t0 + warm < budget_s and stages < max_stages):
            cp.step(case.dt)
            stages += 3
        return case, cp, stages, time.perf_counter() - t0, warm

    rs_cpu = min(rs, 4 if order <= 3 else (3 if order <= 4 else 2))  # a size that certainly fits the budget ...
    case, cp, stages, el, warm = run(rs_cpu, 30)
    rate = case.u0.size * stages / el
    if rs_cpu < rs and 8.0 * case.u0.size * 6 / rate < budget_s:  # (8 x the dofs; warm-up step + one timed step = 6 stages)
        case, cp, stages, el, warm = run(rs_cpu + 1, 45)  # ... and the GPU's own mesh when a warm-up + one step of it fit too (then up to the budget)
        rs_cpu += 1
    ndofs = case.u0.size
    tb = cp.buckets()
    fom = lambda t: 1e-6 * ndofs * stages / t if t > 0 else None
    return {
        "value": 1e-6 * ndofs * stages / el,
        "unit": "MDOFs*RK-stage/s",
        "cores": cp.threads,
        "kind": "port",
        "sample_short": f"periodic-cube -rs {rs_cpu} -o {order} -p 10 -lo 5 -fct 2, {ndofs} dofs, {stages} RK stages in {el:.2f} s, {cp.threads} OpenMP threads",
        "sample": f"oracle/cpu_port.cpp (C++/OpenMP, {cp.threads} threads = the CPUs the container may use (affinity mask capped by the "
                  f"cgroup quota), element batches of {cp.simd_width} per SIMD vector, "
                  f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}): periodic-cube -rs {rs_cpu} "
                  f"-o {order} -p 10 -lo 5 -fct 2, mass solve {MASS_SOLVE[mass_solve][1]}, {ndofs} dofs, {stages} RK stages in {el:.2f} s",
        "buckets_s": {"rhs": tb[0], "inv": tb[1], "lo": tb[2], "fct": tb[3]},
        "fom_reference_style": {"rhs": fom(tb[0]), "inv": fom(tb[1]), "lo": fom(tb[2]), "fct": fom(tb[3]),
                                "total_rhs_lo_fct": fom(tb[0] + tb[2] + tb[3])},
    }


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process group (torch.distributed.run) and
    pass its output and exit code through.  Nothing in this process has touched the GPU (torch is not even imported).
    Watchdog: after $RMH_BENCH_TIMEOUT seconds (default 1500) the children -- the whole process group, never this process
    re-exec'ed -- are terminated and the exit code is 124."""
    import signal

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("RMH_LAUNCH_NONCE", f"{os.getpid()}_{int(time.time())}")  # (shared by every rank of this launch)
    limit = float(os.environ.get("RMH_BENCH_TIMEOUT", "1500"))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {args.gpus} ranks did not finish within {limit:.0f} s -- terminating them (per-rank logs: "
              f"{rank_log_path('*')})", file=sys.stderr, flush=True)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def rank_log_path(rank):
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d if os.path.isdir(d) else os.environ.get("TMPDIR", "/tmp"), f"bench_rank{rank}.log")


_T0 = time.time()


def rank_log(rank, msg):
    """every rank of an N > 1 run keeps its own log (first contact with a multi-GPU node: know where each rank stopped)"""
    try:
        with open(rank_log_path(rank), "a") as f:
            f.write(f"[{time.strftime('%H:%M:%S')} +{time.time() - _T0:7.1f}s pid {os.getpid()}] {msg}\n")
    except OSError:
        pass


def start_rank_watchdog(rank, world):
    """N > 1: a rank stuck in a collective never returns by itself.  After $RMH_BENCH_RANK_TIMEOUT seconds (default 1200)
    the rank says so in its log and on stderr and exits with 124 (os._exit: no exec, no clean-up that could block)."""
    import threading

    limit = float(os.environ.get("RMH_BENCH_RANK_TIMEOUT", "1200"))

    def fire():
        msg = f"rank {rank}/{world}: watchdog -- not finished after {limit:.0f} s, exiting (see {rank_log_path(rank)})"
        rank_log(rank, msg)
        print("bench.py: " + msg, file=sys.stderr, flush=True)
        os._exit(124)

    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


def all_ranks_ok(tag, world, rank, ok, timeout_s=180.0):
    """Every rank learns whether ALL ranks succeeded -- one small file per rank, no torch and no RCCL needed (the step this
    guards is the one that may have failed to reach RCCL).  A rank that alone fell back to the Python loop would sit in
    torch.distributed's rendezvous while the others have printed and left."""
    base = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rmh_bench_{tag}")
    mine = f"{base}.rc{rank}"
    with open(mine + ".tmp", "w") as f:
        f.write("1" if ok else "0")
    os.replace(mine + ".tmp", mine)
    t0, vals = time.time(), []
    while time.time() - t0 < timeout_s:
        vals = []
        for r in range(world):
            try:
                vals.append(open(f"{base}.rc{r}").read().strip())
            except OSError:
                vals.append(None)
        if all(v in ("0", "1") for v in vals):
            break
        time.sleep(0.05)
    if rank == 0:  # rank 0 is the one that prints: it clears ALL the files, after every rank had time to read them
        import atexit

        def _clean():
            time.sleep(1.0)
            for r in range(world):
                try:
                    os.remove(f"{base}.rc{r}")
                except OSError:
                    pass

        atexit.register(_clean)
    return all(v == "1" for v in vals)


class SmiSampler:
    """Shader clock and socket power from rocm-smi while the timed region runs (a host-side subprocess in its own
    thread; best effort: any failure just leaves the list empty)."""

    def __init__(self, period_s=0.1):
        import threading

        self.period_s = period_s
        self.samples, self._stop = [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def allowed():
        """Not under a profiler: its preloaded tool library would be inherited by the child process."""
        env = os.environ
        pre = env.get("LD_PRELOAD", "").lower()
        return not (any(w in pre for w in ("rocprof", "roctracer", "rocprofiler")) or env.get("HSA_TOOLS_LIB")
                    or any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER")) for k in env))

    def _run(self):
        import re
        import shutil

        smi = shutil.which("rocm-smi")
        if not smi:
            return
        smi = os.path.realpath(smi)
        # (the script is started with this interpreter directly: no `env` hop; a plain environment)
        cmd = [sys.executable, smi] if smi.endswith(".py") else [smi]
        env = dict(os.environ)
        while not self._stop.is_set():
            try:
                t = time.perf_counter()
                txt = subprocess.run(cmd + ["--showclocks", "--showpower"], capture_output=True, text=True, timeout=3, env=env).stdout
                sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
                pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
                if sclk and pw:
                    self.samples.append((0.5 * (t + time.perf_counter()), int(sclk.group(1)), float(pw.group(1))))
            except Exception:  # noqa: BLE001
                return
            self._stop.wait(self.period_s)

    def start(self):
        self._thread.start()

    def stop(self, t0, t1):
        self._stop.set()
        self._thread.join(timeout=5)
        inside = [s for s in self.samples if t0 <= s[0] <= t1]
        return {"samples_in_timed_region": len(inside), "sclk_MHz": [s[1] for s in inside], "power_W": [s[2] for s in inside],
                "source": "rocm-smi --showclocks --showpower, sampled from a host thread during the timed region"}


_CASES = {}


def measure(args, lib, order, rs, world, rank, dev, dist, backend, with_counters=True, problem=None, mass_solve=None,
            steps=None, warmup=None, smi_period=None, per_step=False, keep_state=False, lo=None, mesh=None, dt_scale=1.0):
    """Set up one configuration, time K steps after W warm-up steps, return the result fields.
    smi_period: sample rocm-smi from a host thread every so many seconds during the timed region (None: not at all --
    the headline figure is measured without instrumentation); per_step: HIP events between the steps (first / steady)."""
    import torch

    from remhos_amd.case import Case, make_config
    from remhos_amd.stepper import Stepper

    problem = args.problem if problem is None else problem
    mesh = args.mesh if mesh is None else mesh
    lo = args.lo if lo is None else lo
    mass_solve = args.mass_solve if mass_solve is None else mass_solve
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    part = PART[world]
    weak = args.scaling == "weak" and world > 1
    # weak scaling: one -rs block per rank -- the directions that carry two blocks are refined once more
    extra = tuple(1 if (weak and part[d] == 2) else 0 for d in range(3))
    cfg = make_config(mesh, rs, order, problem, -1.0, 0.5, lo_type=lo, part=part, rank=rank, rs_extra=extra, tile_rows=args.tile)
    t0 = time.perf_counter()
    key = (mesh, rs, order, problem, lo, part, rank, extra, args.tile)
    if key not in _CASES:  # (host-side case set-up, 3-4 s at -rs 5: shared by the blocks of one run)
        _CASES.clear()
        _CASES[key] = Case(lib, cfg)
    case = _CASES[key]
    st = Stepper(lib, case, device=dev, dist=dist, fused=not args.unfused, one_kernel=not args.two_kernels,
                 overlap=os.environ.get("RMH_NO_OVERLAP", "0") != "1")  # (escape hatch: exchange, then one launch)
    (rel, ab, mit, jac, fix), mass_tol = MASS_SOLVE[mass_solve]
    st.ctx.set_mass_tol(rel, ab, mit)
    st.ctx.set_mass_completion(jac, fix)
    setup_s = time.perf_counter() - t0
    global_dofs = case.ne_global * case.ndof
    dt = case.dt * dt_scale  # (dt_scale < 1: the same work per step at a smaller step -- the stable-step accuracy legs)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    smi = SmiSampler(smi_period) if (smi_period and world == 1 and SmiSampler.allowed()) else None
    for _ in range(warmup):
        st.step(dt)
    st.ctx.last_cg_iters()  # reset
    st.ctx.enable_timers(True)
    st.ctx.reset_timers()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step else None
    barrier()
    if smi:
        smi.start()
    t0 = time.perf_counter()
    for k in range(steps):
        if evs:
            evs[k].record()  # (the stepper's kernels run on torch's current stream)
        st.step(dt)
    if evs:
        evs[steps].record()
    barrier()
    elapsed = time.perf_counter() - t0
    smi_info = smi.stop(t0, t0 + elapsed) if smi else None
    step_ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(steps)] if evs else None
    tim = st.ctx.timers()
    st.ctx.enable_timers(False)
    cg_iters = st.ctx.last_cg_iters()
    rdev = dev if (dist is not None and dist.get_backend() == "nccl") else "cpu"  # (gloo reduces host tensors)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    stages = 3 * steps
    value = 1e-6 * global_dofs * stages / elapsed
    # sanity of the state after the run: mass conservation and bounds (not timed)
    mass, umax = st.local_mass_and_max()
    state = st.x.clone() if keep_state else None
    if dist is not None:
        red = torch.tensor([mass], dtype=torch.float64, device=rdev)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        mass = float(red[0])
        red = torch.tensor([umax], dtype=torch.float64, device=rdev)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        umax = float(red[0])
    one_kernel = st.one_kernel
    ne_owned, ndof, ne_global, n_lat = case.ne_owned, case.ndof, case.ne_global, list(case.n)
    xinfo = None
    if case.peers:
        _, sn, _, gn = st.ctx.exchange_buffers()
        xinfo = {"transport": {"rccl": "RCCL grouped ncclSend/ncclRecv inside the library (rmh_exchange_begin/_end)",
                               "external": "torch.distributed isend/irecv on the library's segments"}.get(st.transport, st.transport),
                 "ghost_records": "compact (face layer + extrema; extrema only for edge/vertex neighbours)" if st.compact else "whole elements",
                 "neighbour_ranks": len(case.peers), "send_bytes_per_stage_rank0": 8 * sn, "recv_bytes_per_stage_rank0": 8 * gn,
                 "rccl_fallback_reason": getattr(st, "rccl_error", None)}
    st.close()
    del st
    torch.cuda.empty_cache()

    ho_avg_s = tim[0] / stages
    if one_kernel:
        # the dominant kernel is the whole stage: SURVEY 8(d) matrix-free per-dof figure
        ho_bytes = int(stage_alg_bytes_per_dof(order, lo) * ne_owned * ndof)
        kname = f"rmh::ho_kernel2<{order}, {3 if lo in (3, 4) else 1}> (whole RK stage)"
    else:
        ho_bytes = ho_alg_bytes_per_element(order) * ne_owned
        kname = f"rmh::ho_kernel2<{order}, 0>"
    achieved = ho_bytes / ho_avg_s / 1e9
    traffic, fp64, why, traffic_x2 = None, None, "not looked up", None
    if with_counters:
        key = f"{mesh}-rs{rs}-o{order}-n{world}" + ("-stage" if one_kernel else "") + ("-lo4" if lo == 4 else "")
        traffic, wi, why = stored_counters(key, mass_tol, lo)
        traffic_x2 = _X2.get(key) if traffic else None
        if wi:
            # the kernel's real bound: FP64 VALU.  Issued lane-flops = wave64 instructions x 64 lanes x (2 for FMA)
            flops = 64.0 * (2.0 * wi["fma"] + wi["mul"] + wi["add"])
            fp64 = {"bound": "fp64 valu", "flops_per_launch": flops, "achieved": flops / ho_avg_s / 1e12,
                    "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / ho_avg_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                    "fp64_share_of_valu_insts": (wi["fma"] + wi["mul"] + wi["add"]) / wi["all_valu"]}
    lattice = "x".join(str(k) for k in n_lat)
    res = {
        "value": value,
        "ms_per_step": 1e3 * elapsed / steps,
        "config": {
            "workload": f"{mesh} -rs {rs}{'+' + ''.join(str(k) for k in extra) if any(extra) else ''} -o {order} -p {problem} "
                        f"{'remap' if problem >= 10 else 'transport'}, -pa -ho 3 -lo {lo} -fct 2, RK3-SSP"
                        + (" (BASELINE configs[1])" if (mesh, order, problem, lo, rs) == ("periodic-cube", 3, 10, 5, 5) and world == 1 else "")
                        + (" (BASELINE configs[2])" if (mesh, order, problem, lo, rs) == ("periodic-cube", 6, 10, 5, 4) and world == 1 else "")
                        + (" (BASELINE configs[3])" if (mesh, order, problem, lo) == ("periodic-cube", 3, 10, 5) and world > 1 else "")
                        + (" (BASELINE configs[4]'s mesh and order on one GPU)" if (mesh, order, problem, lo, rs) == ("cube01_hex", 4, 10, 5, 5) and world == 1 else "")
                        + f"; {lattice} = {ne_global} hex, {global_dofs} dofs",
            "global_dofs": global_dofs,
            "elements": ne_global,
            "dofs_per_gpu": ne_owned * ndof,
            "partition": "x".join(str(k) for k in part),
            "tile_rows": args.tile,
            "limiter": "reference call sequence" if args.unfused else ("inside the stage kernel" if one_kernel else "fused (LO avg + bounds + ClipScale + RK update)"),
            "dt": dt,
            "dt_rule": "-dt -1: 0.25 h / |v| (remhos.cpp:538-553)" + (f" x {dt_scale:.6g}" if dt_scale != 1.0 else ""),
            "mass_cg_max_iters": cg_iters,
            "mass_solve": mass_solve,
            "mass_tol": mass_tol,
            "final_mass": mass,
            "max_value": umax,
            "setup_s": setup_s,
        },
        "roofline": {
            "kernel": kname,
            "bound": "hbm",          # the roof `achieved / peak / frac` are priced against (the bench contract's model) ...
            "binds": "fp64-valu",    # ... and the resource that actually binds the kernel (roofline_fp64, `of_bound`)
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_x2,
            # which roof: `frac` prices MODEL bytes against the HBM peak.  dram_frac = measured bytes / launch / peak (what the memory
            # system really moves); of_bound = the kernel's tighter bound -- max(measured bytes / sustainable HBM rate, issued FP64
            # flops / vector peak) -- over the measured launch time, i.e. how far the launch is from the roof that limits it
            "dram_frac": (traffic_x2 / ho_avg_s / 1e9 / HBM_PEAK_GBS) if traffic_x2 else None,
            "of_bound": (max(traffic_x2 / (HBM_SUSTAINED_GBS * 1e9), fp64["flops_per_launch"] / (FP64_VALU_PEAK_TFLOPS * 1e12)) / ho_avg_s)
                        if (traffic_x2 and fp64) else None,
            "of_bound_is": f"max(traffic / {HBM_SUSTAINED_GBS:.0f} GB/s, FP64 flops / {FP64_VALU_PEAK_TFLOPS} TFLOP/s) / avg launch time",
            "traffic_cal1771": traffic,
            "traffic_is": "measured HBM bytes per launch: WRITE_SIZE + 2 x FETCH_SIZE -- the gfx950 correction of MI355X_MICROARCH.md, "
                          "which profiles/r04_streaming_traffic.txt confirms for this library's 8-byte-per-lane streams (mass-based "
                          "average, ClipScale: 2 x FETCH_SIZE = the algorithmic read bytes to 4 digits); `traffic_cal1771`: the figure of "
                          "rounds 1-4 (x 1.771, calibrated on the first-generation limiter kernel, whose assumed read bytes were too low)",
            "traffic_source": why,
            "avg_launch_ms": 1e3 * ho_avg_s,
            "alg_bytes_per_launch": ho_bytes,
            "achieved_is": "ALGORITHMIC bytes of SURVEY 8(d) (matrix-free model) / launch time -- a model figure, not measured HBM GB/s; "
                           "the measured HBM bytes per launch are `traffic`",
            "binds_is": "the resource that binds: FP64 vector issue + LDS (the volume geometry is recomputed from the 27 nodes every stage), see "
                        "roofline_fp64 and DESIGN.md 3.1; achieved / peak / frac are the HBM model (`bound`) the bench contract asks for",
        },
        "roofline_fp64": fp64,
        "smi": smi_info,
        "step_ms": step_ms,
        "_state": state,
        "buckets_s": {"ho_rhs_plus_inv_or_stage": tim[0], "lo": tim[2], "fct_or_fused_limiter": tim[3]},
        # the reference's figures of merit (remhos.cpp:1918-1966): 1e-6 * dofs * stages / bucket time of rank 0; its
        # printed total uses T_rhs + T_LO + T_FCT.  Here RHS and INV are one kernel (bucket 0); with the one-kernel
        # stage everything is in bucket 0.  "value" above is the all-inclusive wall-clock figure.
        "fom_reference_style": {
            k: (1e-6 * global_dofs * stages / v if v > 0 else None)
            for k, v in (("rhs_plus_inv", tim[0]), ("lo", tim[2]), ("fct", tim[3]), ("total_rhs_lo_fct", tim[0] + tim[2] + tim[3]))
        },
        "exchange": xinfo,
        "stage_roofline": {
            "alg_bytes_per_dof": stage_alg_bytes_per_dof(order, lo),
            "achieved_GBs": value * 1e6 * stage_alg_bytes_per_dof(order, lo) / 1e9,
            "frac_of_hbm_peak": value * 1e6 * stage_alg_bytes_per_dof(order, lo) / 1e9 / (HBM_PEAK_GBS * world),
        },
    }
    return res


def granular_block(args, lib, world, rank, dev, dist, backend):
    """The reference's own call sequence at the plug-in boundary (remhos.cpp:1692, 1815-1831): CalcHOSolution, CalcLOSolution,
    ComputeElementsMinMax + ComputeBounds, CalcFCTSolution as separate kernels, the RK vector updates by the caller (y.Add: torch; add(a, x, b, y, z): rmhd_axpby) --
    what a binding that only replaces the three solver classes gets -- and the HO kernel + fused limiter.
    The limiter-side kernels stream every E-vector once: each is priced against its algorithmic HBM bytes (8 B x dofs x
    vectors read and written, + the stencil table), timed alone with HIP events on the context's stream."""
    import copy

    import torch

    from remhos_amd.stepper import Stepper

    a2 = copy.copy(args)
    a2.unfused = True
    seq = measure(a2, lib, args.order, args.rs, world, rank, dev, dist, backend, with_counters=False)
    a3 = copy.copy(args)
    a3.two_kernels = True
    two = measure(a3, lib, args.order, args.rs, world, rank, dev, dist, backend, with_counters=False)
    case = next(iter(_CASES.values()))
    st = Stepper(lib, case, device=dev, fused=False)
    (rel, ab, mit, jac, fix), _ = MASS_SOLVE[args.mass_solve]
    st.ctx.set_mass_tol(rel, ab, mit)
    st.ctx.set_mass_completion(jac, fix)
    for _ in range(2):
        st.step(case.dt)
    ctx, u, dt = st.ctx, st.x, case.dt
    n, ne = case.ne_owned * case.ndof, case.ne_owned
    f64 = dict(dtype=torch.float64, device=u.device)
    du_ho, du_lo, du, umin, umax, y = (torch.empty(n, **f64) for _ in range(6))
    xe_min, xe_max = torch.empty(ne, **f64), torch.empty(ne, **f64)
    ctx.setup(st.t)
    ctx.ho_apply(u, du_ho)
    m_ptr = ctx.lumped_mass_ptr()
    vec, sten = 8.0 * n, 4.0 * 27 * ne
    rows = [
        ("elem_minmax_kernel", "ComputeElementsMinMax (remhos_tools.cpp:497-523)", lambda: ctx.elem_minmax(u, xe_min, xe_max), vec + 16.0 * ne),
        ("bounds_kernel", "ComputeBounds (remhos_tools.cpp:432-495)", lambda: ctx.bounds(xe_min, xe_max, umin, umax), 2 * vec + sten + 16.0 * ne),
        ("lo_massavg_kernel", "MassBasedAvg::CalcLOSolution (remhos_lo.cpp:247-324)", lambda: ctx.lo_massavg(u, du_ho, dt, du_lo), 4 * vec),
        ("fct_clipscale_kernel", "ClipScaleSolver::CalcFCTSolution (remhos_fct.cpp:449-541)",
         lambda: ctx.fct_clipscale(u, m_ptr, du_ho, du_lo, umin, umax, dt, du), 7 * vec),
        ("limit_fused_kernel", "LimitMult + RK update in one pass (remhos.cpp:1798-1845)",
         lambda: ctx.limit_fused(u, du_ho, dt, x_base=u, a=0.75, b=0.25, dt_rk=dt, y_out=y), 5 * vec + sten),
    ]
    kern = {}
    reps = 20
    for name, what, fn, nbytes in rows:
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        kern[name] = {"replaces": what, "avg_launch_ms": ms, "alg_bytes_per_launch": nbytes, "bound": "hbm",
                      "achieved": nbytes / (1e9 * ms) * 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (1e6 * ms) / HBM_PEAK_GBS}
    st.close()
    del st
    torch.cuda.empty_cache()
    keep = ("value", "ms_per_step", "buckets_s", "fom_reference_style")
    return {"what": "the same workload through the GRANULAR entry points of the C ABI: the reference's call sequence "
                    "(rmh_ho_apply, rmh_lo_massavg, rmh_elem_minmax, rmh_bounds, rmh_fct_clipscale; RK updates by the caller) "
                    "and HO kernel + fused limiter (rmh_ho_apply, rmh_limit_fused)",
            "unit": "MDOFs*RK-stage/s", "steps": args.steps, "warmup": args.warmup,
            "reference_call_sequence": {k: seq[k] for k in keep},
            "ho_plus_fused_limiter": {k: two[k] for k in keep},
            "ho_kernel": {"kernel": f"rmh::ho_kernel2<{args.order}, 0>", "avg_launch_ms": 1e3 * two["buckets_s"]["ho_rhs_plus_inv_or_stage"] / (3 * args.steps)},
            "streaming_kernels": kern,
            "timing": f"each streaming kernel alone, {reps} launches between two HIP events on the context's stream, after 2 RK steps"}


def config0_block(lib):
    """BASELINE.json configs[0] -- the reference's own CPU-runnable case, 2-D periodic-square transport with RK3 -- through the
    product's harness on the GPU (rmhd_run: C++ case builder, solver classes, dim = 2 kernels): the autotest line
    `-m data/periodic-square.mesh -p 5 -rs 3 -dt 0.004 -tf 0.8 -ho 3 -lo 4 -fct 2` (order 3, 200 steps; autotest/test.sh:51,
    expected autotest/out_baseline.dat:61-64) with its printed mass and maximum as a verdict, and configs[0]'s own order 2.
    A parity block, not a performance claim: 9 216 / 5 184 dofs, a few launches of ~10 us per stage."""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    out = {"what": "BASELINE configs[0] (2D periodic-square transport, RK3, -ho 3 -lo 4 -fct 2) on the GPU through rmhd_run; dim = 2 kernels "
                   "(remhos_amd/csrc/rmh_2d.hpp + the streaming kernels), the reference's call sequence (fused = 0)"}
    ref = {"mass": 0.1623263888, "max": 0.7145371968, "source": "autotest/out_baseline.dat:61-64 (10 significant digits, order 3)"}
    for order in (3, 2):
        res = RmhdResult()
        cfg = make_config("periodic-square", 3, order, 5, 0.004, 0.8, lo_type=4, fused=0)
        if lib.rmhd_run(C.byref(cfg), C.byref(res)) != 0:
            out[f"o{order}"] = {"error": lib.rmhd_last_error().decode()}
            continue
        blk = {"workload": f"periodic-square -p 5 -rs 3 -o {order} -dt 0.004 -tf 0.8 -ho 3 -lo 4 -fct 2", "steps": res.steps,
               "dofs": int(res.global_dofs), "final_mass": res.final_mass, "max_value": res.max_value,
               "value": res.fom_wall, "unit": "MDOFs*RK-stage/s (wall clock of the whole loop)", "wall_s": res.wall}
        if order == 3:
            blk["reference"] = ref
            blk["pass"] = bool(float(f"{res.final_mass:.10g}") == ref["mass"] and float(f"{res.max_value:.10g}") == ref["max"])
        out[f"o{order}"] = blk
    return out


def measure_cpp_loop(args, lib, world, rank, device, comm_file, extra_override=None):
    """N > 1: the C++ stage loop (rmhd_run_partitioned, remhos_amd/csrc/rmh_driver.hip) -- one process per GPU, the
    halo exchange as grouped RCCL send/recv inside the library, no Python between the launches (at N = 8 strong
    scaling a block's stage takes ~0.5 ms).  W warm-up steps, then K timed steps between two barriers (one-double
    all-reduces) + device synchronisations; wall = max over ranks.  comm_file = None: all blocks in THIS process on one
    GPU, device copies instead of RCCL (RMH_BENCH_ONE_GPU=1, validation of this code path on a 1-GPU box)."""
    import ctypes as C

    from remhos_amd.case import RmhdResult, make_config

    part = PART[world]
    weak = args.scaling == "weak" and world > 1
    extra = tuple(1 if (weak and part[d] == 2) else 0 for d in range(3))
    if extra_override is not None:  # (the one-block reference run of a weak-scaling lattice: mass_check of the N > 1 line)
        extra = tuple(extra_override)
    (rel, ab, mit, jac, fix), mass_tol = MASS_SOLVE[args.mass_solve]
    if args.mass_solve not in ("pa", "exact"):
        raise SystemExit("--gpus N > 1 runs the C++ loop: --mass-solve pa or exact")
    cfg = make_config(args.mesh, args.rs, args.order, args.problem, -1.0, 0.5, max_steps=args.warmup + args.steps, lo_type=args.lo,
                      part=part, rank=rank, rs_extra=extra, pa=1 if args.mass_solve == "pa" else 0, warmup_steps=args.warmup,
                      tile_rows=args.tile)
    res = RmhdResult()
    t0 = time.perf_counter()
    rc = lib.rmhd_run_partitioned(C.byref(cfg), comm_file.encode() if comm_file else None, device, C.byref(res))
    if rc != 0:
        # (e.g. RCCL not reachable through dlopen: the same on every rank -- the caller falls back to the Python loop)
        print(f"rank {rank}: rmhd_run_partitioned: {lib.rmhd_last_error().decode()}", file=sys.stderr, flush=True)
        return None
    total_s = time.perf_counter() - t0
    order, D = args.order, args.order + 1
    stages = res.timed_stages
    ne_global = res.global_dofs // D**3
    ho_avg_s = res.t_rhs / stages if res.t_rhs > 0 else res.wall / stages  # (interior + halo-shell launch of a stage, HIP events on the context's stream, max over ranks; RMH_DRIVER_TIMERS=0: the wall clock)
    ho_bytes = int(stage_alg_bytes_per_dof(order) * res.global_dofs / world)
    achieved = ho_bytes / ho_avg_s / 1e9
    return {
        "value": 1e-6 * res.global_dofs * stages / res.wall,
        "ms_per_step": 1e3 * res.wall / args.steps,
        "config": {
            "workload": f"{args.mesh} -rs {args.rs}{'+' + ''.join(str(k) for k in extra) if any(extra) else ''} -o {order} -p {args.problem} "
                        f"{'remap' if args.problem >= 10 else 'transport'}, -pa -ho 3 -lo {args.lo} -fct 2, RK3-SSP"
                        + (" (BASELINE configs[3])" if (args.mesh, order, args.problem, args.lo) == ("periodic-cube", 3, 10, 5) else "")
                        + f"; {ne_global} hex, {res.global_dofs} dofs",
            "global_dofs": res.global_dofs, "elements": ne_global, "dofs_per_gpu": res.global_dofs // world,
            "partition": "x".join(str(k) for k in part), "tile_rows": args.tile, "limiter": "inside the stage kernel", "dt": res.dt,
            "mass_cg_max_iters": res.cg_iters_max, "mass_solve": args.mass_solve, "mass_tol": mass_tol,
            "final_mass": res.final_mass, "max_value": res.max_value, "mass_loss": res.mass_loss,
            "stage_loop": "C++ (rmhd_run_partitioned)", "setup_and_run_s": total_s, "rs_extra": list(extra),
            "comm_ranks": res.comm_ranks if comm_file else None,  # ncclCommCount of the library's communicator (None: no RCCL, one-GPU validation mode)
        },
        "roofline": {
            "kernel": f"rmh::ho_kernel2<{order}, {3 if args.lo in (3, 4) else 1}> (whole RK stage; interior + halo-shell launch)",
            "bound": "hbm", "binds": "fp64-valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "traffic_source": "PMC passes are single-GPU (profiles/)", "avg_launch_ms": 1e3 * ho_avg_s,
            "alg_bytes_per_launch": ho_bytes,
            "achieved_is": "ALGORITHMIC bytes of SURVEY 8(d) (matrix-free model) per GPU and stage / stage kernel time (max over ranks)",
        },
        # (the bucket FOMs of the C++ loop are SAMPLED: events around every timer_every-th step, scaled; `value` is the exact wall clock)
        "fom_reference_style": {"rhs_plus_inv": res.fom_rhs, "total_rhs_lo_fct": res.fom, "sampled_every_nth_step": res.timer_every,
                                "sampled_steps": res.timer_steps},
        "exchange": {
            "transport": {1: "RCCL grouped ncclSend/ncclRecv inside the library (rmh_exchange_begin/_end), exchange stream with priority",
                          2: "same-process device copies (all blocks on one GPU: validation mode)"}.get(res.transport, "none"),
            "ghost_records": "compact (face layer + extrema; extrema only for edge/vertex neighbours)",
            "neighbour_ranks": res.n_peers, "send_bytes_per_stage_rank0": res.send_bytes_per_stage,
            "recv_bytes_per_stage_rank0": res.recv_bytes_per_stage,
        },
        "stage_roofline": {
            "alg_bytes_per_dof": stage_alg_bytes_per_dof(order),
            "achieved_GBs": 1e-6 * res.global_dofs * stages / res.wall * 1e6 * stage_alg_bytes_per_dof(order) / 1e9,
            "frac_of_hbm_peak": 1e-6 * res.global_dofs * stages / res.wall * 1e6 * stage_alg_bytes_per_dof(order) / 1e9 / (HBM_PEAK_GBS * world),
        },
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
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: one -rs block per GPU (weak, default) or the same -rs mesh partitioned (strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-p6", action="store_true", help="skip the p = 6 block (BASELINE configs[2]) of the N = 1 line")
    ap.add_argument("--unfused", action="store_true", help="reference call sequence instead of the fused limiter")
    ap.add_argument("--mass-solve", choices=("pa", "exact", "reference"), default="pa",
                    help="local mass solve: pa = DGMassInverse's abs 1e-8 (remhos_ho.cpp:79-80) + Jacobi step + constant mode "
                         "(default); exact = converged to rel 1e-14; reference = the literal abs 1e-8 rule, no completion")
    ap.add_argument("--smi", action="store_true", help="also sample rocm-smi (clock, power) during the HEADLINE timed region "
                    "(off: the figure is measured without instrumentation; the `sustained` block always samples)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: no transport / sustained / mass_check blocks -- every launch of the run then uses the timed "
                         "configuration (the command the rocprofv3 summaries under profiles/ are taken from)")
    ap.add_argument("--no-smi", action="store_true", help=argparse.SUPPRESS)  # (accepted for old recipes: sampling is opt-in now)
    ap.add_argument("--sustained-steps", type=int, default=200)
    ap.add_argument("--two-kernels", action="store_true", help="HO kernel + fused limiter kernel instead of the one-kernel stage")
    ap.add_argument("--tile", type=int, default=0,
                    help="element numbering of the case builder (rmhd_config.tile_rows): y-strips of this many lattice rows, z before y "
                         "inside a strip; 0 = lattice order (default).  The same mesh and results element for element.  Measured "
                         "(profiles/r05_face_table.txt): 4 rows cut the stage kernel's HBM reads by 12 %% and cost 2.7 %% of its rate")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="N > 1, weak scaling: do not also run the strong-scaling leg (the same -rs mesh partitioned, BASELINE configs[3])")
    ap.add_argument("--cpp-loop", action="store_true",
                    help="N = 1: also time the C++ stage loop of the N > 1 runs (rmhd_run_partitioned) on the 1 x 1 x 1 partition "
                         "(block `cpp_loop`; on by default in the full default run)")
    ap.add_argument("--py-loop", action="store_true",
                    help="N > 1: drive the stages from Python (remhos_amd/stepper.py over torch.distributed) instead of the C++ loop")
    args = ap.parse_args()
    if args.gpus not in PART:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(args))  # (before torch is imported: this process never touches the GPU)
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU")

    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver

    one_gpu = os.environ.get("RMH_BENCH_ONE_GPU", "0") == "1"
    if args.gpus > 1 and not args.py_loop:
        # the C++ loop: nothing of torch is needed (RCCL is reached by the library itself); one rank per GPU as launched,
        # the ncclUniqueId through a file named after this launch's MASTER_PORT
        start_rank_watchdog(rank, world)
        rank_log(rank, f"start: rank {rank}/{world} local_rank {local_rank} scaling {args.scaling} args {sys.argv[1:]}")
        lib = bind_driver(load_library())
        rank_log(rank, "library loaded")
        out, strong = None, None
        if one_gpu:
            if rank == 0:
                out = measure_cpp_loop(args, lib, world, 0, 0, None)
                if args.scaling == "weak" and not args.no_strong_leg:
                    a2 = copy.copy(args)
                    a2.scaling = "strong"
                    strong = measure_cpp_loop(a2, lib, world, 0, 0, None)
        else:
            port = os.environ.get("MASTER_PORT", "0")
            # the launch's tag: the launcher's port, and its run id / the agent's pid so that a stale file of an earlier launch
            # on the same port is never taken for this one's (rmh_driver.hip: read_or_write_id)
            # -- built only from what EVERY rank shares: the agent's pid only when one agent started all ranks of the job
            one_agent = os.environ.get("LOCAL_WORLD_SIZE") == str(world)
            tag = (f"{port}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.environ.get('RMH_LAUNCH_NONCE', '')}"
                   f"_{os.getppid() if one_agent else 0}")
            os.environ.setdefault("RMH_COMM_NONCE", str(int(hashlib.sha256(tag.encode()).hexdigest()[:12], 16)))
            idf = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rmh_bench_{port}")
            rank_log(rank, f"C++ loop ({args.scaling}): rmhd_run_partitioned, id file {idf}.id")
            out = measure_cpp_loop(args, lib, world, rank, local_rank, idf + ".id")
            rank_log(rank, f"C++ loop returned: {'ok' if out is not None else 'FAILED: ' + lib.rmhd_last_error().decode()}")
            # decided collectively: either every rank's C++ loop succeeded, or ALL ranks take the Python loop
            if not all_ranks_ok(tag.replace("/", "_"), world, rank, out is not None):
                rank_log(rank, "not every rank succeeded: all ranks take the Python loop")
                out = None
            if out is not None and args.scaling == "weak" and not args.no_strong_leg:
                # BASELINE configs[3] in the same line: the SAME -rs mesh box-partitioned over the ranks (a second communicator
                # through its own id file; skipped as a whole unless every rank got through it)
                a2 = copy.copy(args)
                a2.scaling = "strong"
                rank_log(rank, "C++ loop (strong leg)")
                strong = measure_cpp_loop(a2, lib, world, rank, local_rank, idf + "_strong.id")
                rank_log(rank, f"strong leg returned: {'ok' if strong is not None else 'FAILED: ' + lib.rmhd_last_error().decode()}")
                if not all_ranks_ok(tag.replace("/", "_") + "_strong", world, rank, strong is not None):
                    strong = None
        if out is None and not one_gpu:
            args.py_loop = True  # best effort: the stages driven from Python over torch.distributed (nccl)
        elif rank == 0 and out is not None:
            line = {"metric": "MDOFs*RK-stage/s, 3D hex remap", "value": out["value"], "unit": "MDOFs*RK-stage/s", "n_gpus": args.gpus,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": out["ms_per_step"], "higher_is_better": True,
                    "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                    "rccl_ranks": out["config"].get("comm_ranks")}
            line.update({k: v for k, v in out.items() if k not in ("value", "ms_per_step")})
            # mass_check of an N > 1 line: the result of a partitioned run does not depend on the partition (tests/test_dist_gloo.py,
            # tests/test_gpu_exchange.py: bit-identical to one block), so rank 0 repeats a leg as ONE block with the same C++ loop and
            # compares the final masses -- the strong leg's mesh (the -rs mesh: always fits one GPU); the weak leg's too where it is
            # small enough to be set up on one GPU in seconds (validation runs; a full -rs 5 block per rank is not repeated)
            checks = {}
            for leg_name, leg, leg_args in (("strong", strong, None), (args.scaling, out, args)):
                if leg is None or leg_name in checks or leg["config"]["global_dofs"] > 80e6:
                    continue
                a1 = copy.copy(args)
                a1.scaling = "strong"
                ref = measure_cpp_loop(a1, lib, 1, 0, 0 if one_gpu else local_rank, None, extra_override=leg["config"]["rs_extra"])
                if ref is None:
                    continue
                m1, mn = ref["config"]["final_mass"], leg["config"]["final_mass"]
                checks[leg_name] = {"one_block_final_mass": m1, "final_mass": mn, "mass_rel_dev": (mn - m1) / m1, "identical": bool(mn == m1),
                                    "pass": bool(abs(mn - m1) <= 1e-12 * abs(m1))}
            if checks:
                first = checks.get(args.scaling) or next(iter(checks.values()))
                line["mass_check"] = {"mass_rel_dev": first["mass_rel_dev"], "field_max_dev": None, "pass": all(c["pass"] for c in checks.values()),
                                      "against": "the same leg run as ONE block by the same C++ loop on rank 0 (final mass; partition independence)",
                                      "legs": checks}
            if strong is not None:
                line["strong_detail"] = strong
                line["strong"] = {"workload": _short_workload(strong["config"]["workload"]), "value": strong["value"],
                                  "ms_per_step": strong["ms_per_step"], "global_dofs": strong["config"]["global_dofs"],
                                  "avg_launch_ms": strong["roofline"]["avg_launch_ms"], "frac": strong["roofline"]["frac"],
                                  "final_mass": strong["config"]["final_mass"], "rccl_ranks": strong["config"].get("comm_ranks")}
            emit(line, args.gpus)
        rank_log(rank, "done" if not args.py_loop else "falling back to the Python loop")
        if not args.py_loop:
            return

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the remhos_amd hot path has no CPU fallback")
    # RMH_BENCH_ONE_GPU=1 (validation aid for boxes with a single GPU, NOT a benchmark): every rank on cuda:0, gloo
    # process group, the exchange through torch.distributed on the library's segments -- exercises the multi-rank code
    # path of this file (weak-scaling lattices, reductions, the JSON line) where RCCL refuses two ranks on one device
    if one_gpu:
        local_rank = 0
        os.environ["RMH_SYNC_EXCHANGE"] = "1"
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist, backend = None, None
    if world > 1:
        import torch.distributed as dist

        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))
        backend = dist.get_backend()

    lib = bind_driver(load_library())
    extras = world == 1 and not args.no_extras
    main_res = measure(args, lib, args.order, args.rs, world, rank, dev, dist, backend, keep_state=extras,
                       smi_period=0.05 if args.smi else None)
    default_case = (args.order, args.mesh, args.problem) == (3, "periodic-cube", 10)

    def verdict(mass_dev, twin_dev):
        """the printed criterion as a boolean: |mass_rel_dev| <= 1e-12 (BASELINE.json north_star), or the round-off twin's
        drift where that is larger (two converged runs that differ by rounding errors only cannot agree better)"""
        return bool(abs(mass_dev) <= max(1e-12, abs(twin_dev)))

    def compare(runs, solve):
        """mass / field deviation of run `solve` and of the round-off twin from the converged run"""
        m_ex = runs["exact"]["config"]["final_mass"]
        devs = lambda k: ((runs[k]["config"]["final_mass"] - m_ex) / m_ex, float((runs[k]["_state"] - runs["exact"]["_state"]).abs().max()))
        (md, fd), (tm, tf) = devs(solve), devs("roundoff-twin")
        return {"mass_rel_dev": md, "field_max_dev": fd, "roundoff_twin_mass_rel_dev": tm, "roundoff_twin_field_max_dev": tf,
                "pass": verdict(md, tm)}

    def mass_check(res, order, rs, mesh=None):
        """Untimed accuracy leg: the same W + K steps with the local solve converged (rel 1e-14) -- `mass_rel_dev` and
        `field_max_dev` of the timed run against it -- and with a round-off twin of the converged run (converged +
        constant mode: differs from it by rounding errors only), whose drift is the floor of that comparison.  `pass` is the
        criterion evaluated.  At p >= 5 the reference's step (-dt -1: p-independent, remhos.cpp:538-553) amplifies a rounding
        error by orders of magnitude within the run -- the CPU oracle does the same, tests/test_oracle_growth.py -- so that
        leg is labelled `chaotic_dt` and a second leg, `stable_dt`, repeats the comparison over 25 steps at dt / (2 p + 1)
        (the same work per step), where two runs stay together and the field deviation is meaningful too."""
        kw = dict(with_counters=False, keep_state=True, mesh=mesh)
        runs = {"timed": res, "exact": measure(args, lib, order, rs, world, rank, dev, dist, backend, mass_solve="exact", **kw),
                "roundoff-twin": measure(args, lib, order, rs, world, rank, dev, dist, backend, mass_solve="roundoff-twin", **kw)}
        out = {"against": f"the same {args.warmup} + {args.steps} steps with the local mass solve converged to rel 1e-14 "
                          f"({runs['exact']['config']['mass_cg_max_iters']} PCG iterations at most; timed run: {res['config']['mass_cg_max_iters']})",
               "converged_final_mass": runs["exact"]["config"]["final_mass"], "converged_value": runs["exact"]["value"],
               **compare(runs, "timed"),
               "criterion": "pass = |mass_rel_dev| <= max(1e-12, |roundoff_twin_mass_rel_dev|): 1e-12 is BASELINE.json north_star's bound; two "
                            "converged runs that differ by rounding errors only (the twin) cannot agree better than their own drift",
               "chaotic_dt": order >= 5,
               "dt_regime": ("the reference's step -dt -1 is beyond the stability limit of the unlimited HO scheme at this order: the run amplifies a "
                             "rounding error ~10x per step (also in the CPU oracle, tests/test_oracle_growth.py); field_max_dev is not "
                             "meaningful after ~8 steps, see stable_dt") if order >= 5 else "stable: two runs a rounding error apart stay together"}
        res.pop("_state")
        for r in runs.values():
            r.pop("_state", None)
        del runs
        torch.cuda.empty_cache()
        # the same comparison over the first 5 steps only: before a run that amplifies perturbations (p = 6) has
        # decorrelated, i.e. the solver's own contribution
        sh = {k: measure(args, lib, order, rs, world, rank, dev, dist, backend, mass_solve=k, steps=5, warmup=0, **kw)
              for k in ("exact", args.mass_solve, "roundoff-twin")}
        out["first_5_steps"] = compare(sh, args.mass_solve)
        del sh
        torch.cuda.empty_cache()
        if order >= 5:
            sc = 1.0 / (2 * order + 1)
            st = {k: measure(args, lib, order, rs, world, rank, dev, dist, backend, mass_solve=k, steps=25, warmup=0, dt_scale=sc, **kw)
                  for k in ("exact", args.mass_solve, "roundoff-twin")}
            out["stable_dt"] = {"steps": 25, "dt": st["exact"]["config"]["dt"], "dt_rule": st["exact"]["config"]["dt_rule"],
                                **compare(st, args.mass_solve),
                                "field_criterion": "field_max_dev bounded by the per-stage tolerance of the element-local solve at this order "
                                                   "times the steps (tests/test_gpu_high_order_runs.py): 25 x 1e-7 at p = 6",
                                "value_at_this_dt": st[args.mass_solve]["value"],
                                "cg_iters": st[args.mass_solve]["config"]["mass_cg_max_iters"]}
            out["stable_dt"]["pass"] = bool(out["stable_dt"]["pass"] and out["stable_dt"]["field_max_dev"] < 25 * {5: 5e-9, 6: 1e-7}[order])
            del st
            torch.cuda.empty_cache()
        return out

    if extras and args.mass_solve != "exact":
        main_res["mass_check"] = mass_check(main_res, args.order, args.rs)
    main_res.pop("_state", None)
    p6, transport, sustained, cube01_p4 = None, None, None, None
    if world == 1 and default_case and not args.no_p6:
        p6 = measure(args, lib, 6, 4, world, rank, dev, dist, backend, keep_state=extras)
        if extras and args.mass_solve != "exact":
            p6["mass_check"] = mass_check(p6, 6, 4)
        p6.pop("_state", None)
    if extras and default_case:
        # BASELINE configs[4]'s mesh and order on one GPU: cube01_hex -rs 5 -o 4 (32.8 M dofs), the moving curvilinear mesh with
        # boundaries -- per-stage operator re-setup is inherent (the geometry is recomputed from the nodes in every stage kernel)
        cube01_p4 = measure(args, lib, 4, 5, world, rank, dev, dist, backend, with_counters=False, keep_state=args.mass_solve != "exact",
                            mesh="cube01_hex")
        if args.mass_solve != "exact":
            cube01_p4["mass_check"] = mass_check(cube01_p4, 4, 5, mesh="cube01_hex")
        cube01_p4.pop("_state", None)
        # the pinned periodic behaviour: transport -p 0 on the same mesh -- true periodic fluxes across the seam
        transport = measure(args, lib, args.order, args.rs, world, rank, dev, dist, backend, with_counters=False, problem=0)
        transport.pop("_state", None)
        # sustained: the headline workload for >= 2 s, with clock / power samples and the first-step / steady split
        sus = measure(args, lib, args.order, args.rs, world, rank, dev, dist, backend, with_counters=False,
                      steps=args.sustained_steps, smi_period=0.1, per_step=True)
        ms = sus["step_ms"]
        srt = sorted(ms)
        sustained = {"steps": args.sustained_steps, "warmup": args.warmup, "value": sus["value"], "unit": "MDOFs*RK-stage/s",
                     "ms_per_step": sus["ms_per_step"], "first_step_ms": ms[0], "median_step_ms": srt[len(srt) // 2],
                     "p05_step_ms": srt[len(srt) // 20], "p95_step_ms": srt[(19 * len(srt)) // 20],
                     "last_10_steps_mean_ms": sum(ms[-10:]) / 10.0, "avg_launch_ms": sus["roofline"]["avg_launch_ms"],
                     "smi": sus["smi"], "workload": sus["config"]["workload"],
                     "note": "the mesh keeps moving: later stages of the remap need the same work per element; a clock or power limit "
                             "would show as step times rising with the sample's power at the cap and sclk falling"}

    granular = None
    if extras and default_case and args.lo == 5 and world == 1 and not args.unfused and not args.two_kernels:
        granular = granular_block(args, lib, world, rank, dev, dist, backend)

    config0 = config0_block(lib) if (extras and default_case and world == 1) else None

    cpp_loop = None
    if world == 1 and rank == 0 and (args.cpp_loop or (extras and default_case)) and args.mass_solve in ("pa", "exact"):
        # the stage loop `--gpus N > 1` runs (rmhd_run_partitioned, C++) on the 1 x 1 x 1 partition of the same workload: the N = 1 point
        # of a scaling curve measured by ONE driver end to end (the headline above is the Python Stepper's loop around the same kernel)
        c1 = measure_cpp_loop(args, lib, 1, 0, local_rank, None)
        if c1 is not None:
            cpp_loop = {"value": c1["value"], "ms_per_step": c1["ms_per_step"], "avg_launch_ms": c1["roofline"]["avg_launch_ms"],
                        "final_mass": c1["config"]["final_mass"], "stage_loop": c1["config"]["stage_loop"],
                        "vs_python_loop": c1["value"] / main_res["value"],
                        # the two loops launch the same kernels in the same order on the same data: the same final mass
                        "pass": bool(abs(c1["config"]["final_mass"] - main_res["config"]["final_mass"]) <= 1e-13 * abs(main_res["config"]["final_mass"]))}

    lo4 = None
    if extras and default_case and args.lo == 5:
        # the LO solver BASELINE.json's north_star names: subcell residual distribution (-lo 4) in the one-kernel stage
        lo4 = {}
        for name, (o4, rs4) in (("p3", (args.order, args.rs)), ("p6", (6, 4))):
            r4 = measure(args, lib, o4, rs4, world, rank, dev, dist, backend, lo=4)
            lo4[name] = {"value": r4["value"], "unit": "MDOFs*RK-stage/s", "ms_per_step": r4["ms_per_step"], "steps": args.steps,
                         "warmup": args.warmup, "avg_launch_ms": r4["roofline"]["avg_launch_ms"], "kernel": r4["roofline"]["kernel"],
                         "roofline": r4["roofline"], "roofline_fp64": r4["roofline_fp64"],
                         "alg_bytes_per_dof": stage_alg_bytes_per_dof(o4, 4),
                         "alg_bytes_model": "SURVEY 8(d) matrix-free stage + the RD solver's own inputs (3 p^3 midpoint and 3 D^3 node velocities of the sub-mesh)",
                         "workload": r4["config"]["workload"],
                         "final_mass": r4["config"]["final_mass"], "mass_cg_max_iters": r4["config"]["mass_cg_max_iters"]}
        _CASES.clear()

    if rank == 0:
        out = {
            "metric": "MDOFs*RK-stage/s, 3D hex remap",
            "value": main_res["value"],
            "unit": "MDOFs*RK-stage/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
        }
        out.update({k: v for k, v in main_res.items() if k not in ("value", "ms_per_step", "step_ms")})
        if world > 1:
            out["rccl_ranks"] = dist.get_world_size()
            out["backend"] = backend
        if p6 is not None:
            p6.pop("step_ms", None)
            out["p6"] = {"metric": out["metric"], "unit": out["unit"], "steps": args.steps, "warmup": args.warmup, **p6}
        if transport is not None:
            keep = ("value", "ms_per_step")
            out["transport"] = {"metric": "MDOFs*RK-stage/s, 3D hex transport", "unit": out["unit"], "steps": args.steps, "warmup": args.warmup,
                                **{k: transport[k] for k in keep}, "config": transport["config"],
                                "avg_launch_ms": transport["roofline"]["avg_launch_ms"], "roofline_hbm_model_frac": transport["roofline"]["frac"]}
        if cube01_p4 is not None:
            cube01_p4.pop("step_ms", None)
            out["cube01_p4"] = {"metric": out["metric"], "unit": out["unit"], "steps": args.steps, "warmup": args.warmup, **cube01_p4}
        if sustained is not None:
            out["sustained"] = sustained
        if lo4 is not None:
            out["lo4"] = lo4
        if granular is not None:
            out["granular"] = granular
        if config0 is not None:
            out["configs0_2d"] = config0
        if cpp_loop is not None:
            out["cpp_loop"] = cpp_loop
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lib, args.order, args.rs, args.mass_solve)
        emit(out, args.gpus)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
