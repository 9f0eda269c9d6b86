"""bench.py's final stdout line is a machine record: it must stay small enough for the driver's stdout tail and parse.

Round 4's line had grown to 22 KB (prose inside every roofline block) and the driver record ended with `parsed: null`.
The canned result is that run's full output (tests/golden/bench_result_r04.json = gpurun_out/bench_drv.json of round 4)."""
import json
import os

import bench

HERE = os.path.dirname(os.path.abspath(__file__))


def canned():
    with open(os.path.join(HERE, "golden", "bench_result_r04.json")) as f:
        return json.load(f)


def test_compact_line_is_small_and_round_trips():
    full = canned()
    assert len(json.dumps(full)) > 20000  # the thing that broke the driver's parse
    txt = bench.compact_line(full)
    assert "\n" not in txt and len(txt) < bench.COMPACT_LIMIT == 6144
    line = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("periodic-cube -rs 5 -o 3 -p 10") and "configs[1]" in line["config"]["workload"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert abs(line["value"] - full["value"]) < 1e-6 * full["value"]
    # the masses keep the digits they are compared on
    assert abs(line["config"]["final_mass"] - full["config"]["final_mass"]) < 1e-12
    assert line["p6"]["pass"] is False and line["p6"]["mass_check"]["stable_dt"]["pass"] is True
    assert line["configs0_2d"]["o3"]["pass"] is True


def test_compact_line_has_no_prose():
    line = json.loads(bench.compact_line(canned()))

    def strings(x, path=""):
        if isinstance(x, dict):
            for k, v in x.items():
                yield from strings(v, f"{path}.{k}")
        elif isinstance(x, str):
            yield path, x

    for path, s in strings(line):
        assert len(s) <= 160, (path, s)


def test_compact_line_sheds_blocks_rather_than_outgrow_the_limit():
    full = canned()
    full["lo4"] = {f"p{k}": full["lo4"]["p3"] for k in range(40)}  # an absurdly wide block
    txt = bench.compact_line(full)
    assert len(txt) < bench.COMPACT_LIMIT
    line = json.loads(txt)
    assert "roofline" in line and "cpu_baseline" in line and "lo4" not in line


def test_multi_gpu_line_is_compact_too():
    full = canned()
    full.update({"n_gpus": 8, "rccl_ranks": 8, "scaling": "weak",
                 "exchange": {"transport": "RCCL grouped ncclSend/ncclRecv inside the library", "neighbour_ranks": 7,
                              "send_bytes_per_stage_rank0": 123456, "recv_bytes_per_stage_rank0": 123456}})
    for k in ("p6", "lo4", "granular", "sustained", "configs0_2d", "cpu_baseline"):
        full.pop(k)
    line = json.loads(bench.compact_line(full, "bench_detail_n8.json"))
    assert line["rccl_ranks"] == 8 and line["exchange"]["neighbour_ranks"] == 7 and line["detail"] == "bench_detail_n8.json"


def test_compact_and_detail_agree_on_what_bounds_the_kernel():
    """the compact record copies `bound` / `binds` from the full result (advisor, round 5: the two outputs contradicted each other):
    `bound` is the roof the contract's achieved / peak / frac are priced against, `binds` the resource that limits the kernel"""
    full = canned()
    r = dict(full["roofline"])
    r.pop("model", None)
    r.update(bound="hbm", binds="fp64-valu", dram_frac=0.25, of_bound=0.36)  # this round's keys
    full["roofline"] = r
    line = json.loads(bench.compact_line(full))
    assert line["roofline"]["bound"] == r["bound"] == "hbm" and line["roofline"]["binds"] == r["binds"] == "fp64-valu"
    assert line["roofline"]["dram_frac"] == 0.25 and line["roofline"]["of_bound"] == 0.36
    # a result of the earlier rounds (model / bound) reads the same way
    old = json.loads(bench.compact_line(canned()))
    assert old["roofline"]["bound"] == "hbm" and old["roofline"]["binds"] == "fp64-valu"


def test_compact_line_never_exceeds_the_limit(capsys):
    """after the optional blocks are gone the record is cut to the contract's fields -- and says so on stderr"""
    full = canned()
    full["strong"] = {f"leg{k}": {"value": 1.0, "note": "x" * 150} for k in range(60)}
    full["cpu_baseline"]["sample_short"] = "y" * 5000
    full["lo4"]["empty"] = None  # (an empty sub-block must not raise)
    txt = bench.compact_line(full)
    assert len(txt) < bench.COMPACT_LIMIT
    line = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert "cutting it to the contract's fields" in capsys.readouterr().err


def test_committed_traffic_entries_belong_to_the_committed_kernel_sources():
    """bench.py reports `roofline.traffic` / `dram_frac` / `of_bound` only from PMC entries measured on the kernel sources it runs
    (profiles/traffic_ho_kernel.json is keyed by their hash): a kernel change without a re-measured profile would silently drop them."""
    have = bench.kernel_source_hash()
    entries = json.load(open(os.path.join(os.path.dirname(HERE), "profiles", "traffic_ho_kernel.json")))
    stage = {k: v for k, v in entries.items() if isinstance(v, dict) and "kernel_src_sha" in v and "-stage" in k}
    assert stage, "no stage entries"
    for k, v in stage.items():
        assert v["kernel_src_sha"] == have, (k, v["kernel_src_sha"], have)
