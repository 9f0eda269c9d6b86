"""First contact with N > 1 as a TESTED COMMAND: `python bench.py --gpus N` exactly as the driver starts it without a launcher --
bench.py's self-launcher, torch.distributed.run, N rank processes, the C++ stage loop rmhd_run_partitioned, the weak leg and the
strong leg, the one-block mass check, the compact record -- on a box with ONE GPU (RMH_BENCH_ONE_GPU=1: every block lives in rank
0's process on that GPU and the halo records move by device copies instead of RCCL; the partition, halo-first element order, pack
kernels, split launches, reductions and the record are the ones of a real node).  What is checked is the plumbing the first SCALE
sweep depends on: exit codes, the record's fields, the per-rank logs, and the watchdog (exit code 124).  No rate is asserted."""
import glob
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(n, extra_env=None, args=("--rs", "2", "--steps", "2", "--warmup", "1"), timeout=900):
    env = dict(os.environ, RMH_BENCH_ONE_GPU="1", **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "bench_rank*.log")):
        os.remove(f)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *args], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("n,partition", [(2, "2x1x1"), (8, "2x2x2")])
def test_bench_gpus_n_on_one_gpu(n, partition):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    p = run_bench(n)
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(p.stdout.strip().splitlines()[-1]) < 6144
    assert line["n_gpus"] == n and line["scaling"] == "weak" and line["config"]["partition"] == partition
    assert line["unit"] == "MDOFs*RK-stage/s" and line["value"] > 0 and line["steps"] == 2 and line["warmup"] == 1
    # weak scaling: one -rs 2 block (12^3 elements, p = 3) per rank; the strong leg: the same -rs 2 mesh over the ranks
    block_dofs = 12**3 * 64
    assert line["config"]["global_dofs"] == n * block_dofs
    assert line["strong"]["global_dofs"] == block_dofs and line["strong"]["value"] > 0
    x = line["exchange"]
    assert x["transport"].startswith("same-process") and x["neighbour_ranks"] >= 1 and x["send_bytes_per_stage_rank0"] > 0
    assert line["rccl_ranks"] is None  # (no communicator in the one-GPU mode; a node reports ncclCommCount here)
    # the partitioned result does not depend on the partition: the same final mass as ONE block, both legs
    assert line["mass_check"]["pass"] is True, line["mass_check"]
    detail = json.load(open(os.path.join(ROOT, f"bench_detail_n{n}.json")))
    legs = detail["mass_check"]["legs"]
    # (the fields are bit-identical -- tests/test_gpu_exchange.py; the printed mass is a sum over blocks: equal up to its last bit)
    assert set(legs) == {"strong", "weak"} and all(abs(v["mass_rel_dev"]) <= 5e-16 for v in legs.values()), legs
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["avg_launch_ms"] > 0
    # every rank keeps its own log
    logs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "bench_rank*.log")))
    assert len(logs) == n, logs
    assert "done" in open(logs[0]).read()


def test_bench_watchdog_exit_code():
    """a launch that cannot finish in time is killed as a process group and reports 124 (what `timeout` reports)"""
    p = run_bench(2, extra_env={"RMH_BENCH_TIMEOUT": "1"}, timeout=300)
    assert p.returncode == 124, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    assert "did not finish within 1 s" in p.stderr


def test_bench_gpus_1_prints_the_cpp_loop_too():
    """`--gpus 1 --cpp-loop`: the N = 1 point of the N > 1 driver (rmhd_run_partitioned on the 1 x 1 x 1 partition) beside the
    headline's Python loop -- the same kernels on the same data: the same final mass"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "RMH_BENCH_ONE_GPU"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rs", "3", "--steps", "2", "--warmup", "1", "--no-extras", "--no-p6",
                        "--no-cpu-baseline", "--cpp-loop"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["cpp_loop"]["pass"] is True and line["cpp_loop"]["value"] > 0
