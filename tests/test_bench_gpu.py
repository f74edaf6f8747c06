"""bench.py's output contract: one JSON line with the driver's keys, the `roofline` and `cpu_baseline` objects at N = 1,
and the N > 1 launch path (torch.distributed.run, one rank per GPU) - exercised here with two ranks sharing the one
GPU of the test box over gloo (FFM_BENCH_ONE_DEVICE=1; the throughput it prints is meaningless)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def last_json(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_and_cpu_baseline_fields():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = last_json(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1
    assert j["unit"] == "images/sec" and j["higher_is_better"] is True and j["scaling"] == "weak" and j["dtype"] == "bf16"
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 32 * 1000.0 / j["ms_per_step"]) < 1e-6 * j["value"]
    ro = j["roofline"]
    assert ro["bound"] == "mfma" and ro["unit"] == "TFLOP/s" and ro["peak"] == 2500.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and 0.05 < ro["frac"] < 1.0
    assert ro["traffic"] is None or ro["traffic"] > 1e6
    # GLP_OT_SVLoRA.train(idx) over one client-round (32 steps of 32): the function SURVEY.md section 8(d) names
    t = j["trainer"]
    assert t["steps_per_round"] == 32
    for mode in ("default", "every_32", "sync_per_step"):
        assert t[mode]["images_per_sec_per_client"] > 100 and t[mode]["ms_per_step"] > 0
    assert t["default"]["images_per_sec_per_client"] > 0.8 * t["every_32"]["images_per_sec_per_client"]


def test_two_ranks_launch_path():
    env = dict(os.environ, FFM_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = last_json(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 2 and j["config"]["global_batch"] == 64 and j["config"]["clients"] == 2
    assert j["config"]["loss_finite"] == 1 and j["config"]["fedavg_payload_bytes"] == 741952 * 4
    assert "cpu_baseline" not in j                                    # rank 0 at N = 1 only


def test_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the two ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["FFM_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline",
                        "--no-trainer"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["rccl_ranks"] == 2 and j["config"]["backend"] == "gloo"
    assert j["config"]["fedavg_round_boundary_us"] > 0 and j["config"]["clients"] == 2
