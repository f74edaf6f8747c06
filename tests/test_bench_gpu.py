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
    # an empty event pair costs a few microseconds; net of it the fraction can only be higher
    assert 0.0 < ro["event_pair_floor_us"] < ro["avg_launch_us"] and ro["frac"] < ro["frac_net_of_event_floor"] < 1.0
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


def _run_bench(extra, env=None, launcher=None, timeout=900):
    base = [sys.executable, "bench.py"] if launcher is None else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher), "--master-addr",
         "127.0.0.1", "--master-port", str(29520 + (launcher or 0) + len(extra)), "bench.py"]
    r = subprocess.run(base + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return last_json(r.stdout)


def test_rccl_executes_at_world_one_and_does_not_slow_the_step():
    """torch.distributed.run with ONE rank and the nccl backend (= RCCL on ROCm): the round-boundary all-reduce really runs
    through an RCCL communicator on this box's GPU, and with a communicator alive the three-stream step keeps its time
    (GPU_MAX_HW_QUEUES=8, DESIGN section 5: RCCL's stream must not push a side stream onto the vision chain's queue).
    No --gpus on the launched command: the world size comes from the launcher."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FFM_BENCH_ONE_DEVICE")}
    common = ["--steps", "30", "--warmup", "5", "--no-roofline", "--no-trainer", "--no-cpu-baseline", "--no-secondary"]
    plain = _run_bench(common, env=env)
    assert plain["config"]["rccl_ranks"] == 1 and plain["config"]["backend"] is None
    j = _run_bench(common, env=env, launcher=1)
    assert j["n_gpus"] == 1 and j["config"]["backend"] == "nccl" and j["config"]["rccl_ranks"] == 1
    assert j["config"]["fedavg_round_boundary_us"] > 0 and j["config"]["fedavg_payload_bytes"] == 741952 * 4
    assert j["config"]["loss_finite"] == 1
    # the un-launched line's 30 steps against the launched line's 30 steps + one round boundary (fractions of a ms)
    assert j["ms_per_step"] < 1.05 * plain["ms_per_step"] + 0.05, (j["ms_per_step"], plain["ms_per_step"])


@pytest.mark.parametrize("cfg,unit,buf", [("c4", "volumes/sec", False), ("c5", "images/sec", True)])
def test_other_baseline_configs_launch_at_two_ranks(cfg, unit, buf):
    """`bench.py --config c4 / c5 --gpus 2` (configs[3] 3D OCT r=16, configs[4] RN50 r=8 G=2): the N > 1 lines exist, end
    with the FedAvg exchange of that workload's trainable buffer and, for RN50, the BatchNorm-buffer all-reduce."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["FFM_BENCH_ONE_DEVICE"] = "1"
    j = _run_bench(["--config", cfg, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline"], env=env)
    assert KEYS <= set(j) and j["n_gpus"] == 2 and j["unit"] == unit and j["config"]["clients"] == 2
    assert j["config"]["backend"] == "gloo" and j["config"]["loss_finite"] == 1
    assert j["config"]["fedavg_payload_bytes"] == 4 * j["config"]["trainable_elems"] > 0
    assert ("fedavg_buffer_payload_bytes" in j["config"]) == buf
    if buf:
        assert j["config"]["fedavg_buffer_payload_bytes"] > 4 * 2 * 1000         # running mean + variance of 53 BatchNorms
    assert ("configs[3]" if cfg == "c4" else "configs[4]") in j["config"]["workload"]


def test_rn50_config_at_world_one_over_rccl_has_a_roofline():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FFM_BENCH_ONE_DEVICE")}
    j = _run_bench(["--config", "c5", "--steps", "5", "--warmup", "2"], env=env, launcher=1)
    assert j["config"]["backend"] == "nccl" and j["config"]["fedavg_buffer_payload_bytes"] > 0
    ro = j["roofline"]
    assert ro["bound"] == "mfma" and 0.01 < ro["frac"] < 1.0 and ro["launches_per_step"] > 100
