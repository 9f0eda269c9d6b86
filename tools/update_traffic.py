"""Fold the rocprofv3 --pmc passes of tools/profile.sh / tools/pmc_insts.sh into profiles/traffic_ho_kernel.json, the
file bench.py reads `roofline.traffic` and `roofline_fp64` from.  Every entry is stamped with the hash of the kernel
sources it was measured on (bench.py refuses an entry whose hash differs), the mass tolerance and the LO solver.

    python tools/update_traffic.py <tag> [mass-solve] [round] [lo]   (reads gpurun_out/prof_<tag>/ and gpurun_out/pmc_<tag>/)

FETCH_SIZE on gfx950 under-reports coalesced reads (MI355X_MICROARCH.md: exactly 1/2 for 16-byte-per-lane streams); for
this kernel's 8-byte-per-lane loads the factor was calibrated in round 1 on limit_fused_kernel, whose read bytes are
known exactly: 1.771 (`hbm_bytes_per_launch`).  Round 4's streaming kernels (profiles/r04_streaming_traffic.txt) show the guide's
factor 2 to be exact for these loads too -- the first-generation limiter read less than was assumed -- so bench.py reports
`hbm_bytes_per_launch_x2` as `roofline.traffic` and the calibrated figure as `traffic_cal1771`.  WRITE_SIZE needs no correction."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import MASS_SOLVE, kernel_source_hash, stage_alg_bytes_per_dof  # noqa: E402

FETCH_CAL = 1.771
tag = sys.argv[1]
solve = sys.argv[2] if len(sys.argv) > 2 else "pa"  # --mass-solve of the profiled bench command
rnd = sys.argv[3] if len(sys.argv) > 3 else "r04"
lo = int(sys.argv[4]) if len(sys.argv) > 4 else 5
mode = 3 if lo in (3, 4) else 1


def counters(pattern):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch = counters(f"prof_{tag}/pmc_fetch/**/*counter_collection.csv")
write = counters(f"prof_{tag}/pmc_write/**/*counter_collection.csv")
insts = counters(f"pmc_{tag}/a/**/*counter_collection.csv")
path = os.path.join(ROOT, "profiles", "traffic_ho_kernel.json")
out = json.load(open(path))
for order, rs, ne in ((3, 5, 884736), (6, 4, 110592)):
    kname = next((k for k in fetch if f"ho_kernel2<{order}, {mode}>" in k), None)
    if not kname:
        print("no dispatches of order", order)
        continue
    avg = lambda d, c: sum(d[kname][c]) / len(d[kname][c])
    f_kib, w_kib = avg(fetch, "FETCH_SIZE"), avg(write, "WRITE_SIZE")
    ndof = (order + 1) ** 3
    ent = {
        "kernel": f"ho_kernel2<{order},{mode}> (whole RK stage, -lo {lo})",
        "kernel_src_sha": kernel_source_hash(), "mass_tol": MASS_SOLVE[solve][1], "lo": lo,
        "fetch_size_kib": f_kib, "write_size_kib": w_kib, "fetch_calibration": FETCH_CAL,
        "hbm_bytes_per_launch": int(1024 * (FETCH_CAL * f_kib + w_kib)),
        "hbm_bytes_per_launch_x2": int(1024 * (2.0 * f_kib + w_kib)),  # the guide's FETCH_SIZE correction for gfx950
        "algorithmic_bytes_per_launch": int(stage_alg_bytes_per_dof(order, lo) * ne * ndof),
        "source": f"profiles/{rnd}{'_lo4' if lo == 4 else ''}_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh {tag})",
    }
    if kname in insts and "SQ_INSTS_VALU_FMA_F64" in insts[kname]:
        ent["fp64_wave_insts_per_launch"] = {
            "fma": avg(insts, "SQ_INSTS_VALU_FMA_F64"), "mul": avg(insts, "SQ_INSTS_VALU_MUL_F64"),
            "add": avg(insts, "SQ_INSTS_VALU_ADD_F64"), "all_valu": avg(insts, "SQ_INSTS_VALU"),
            "source": "rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU (tools/pmc_insts.sh), wave64 instructions",
        }
    out[f"periodic-cube-rs{rs}-o{order}-n1-stage" + ("-lo4" if lo == 4 else "")] = ent
    print(order, ent["hbm_bytes_per_launch"] / 1e9, "GB per launch measured,", ent["algorithmic_bytes_per_launch"] / 1e9, "GB algorithmic")
json.dump(out, open(path, "w"), indent=1)
