#!/usr/bin/env python3
"""Benchmark of the FairLoRA local-training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

The timed region is ONE GLP_OT_SVLoRA.train(idx) call over a local epoch of K
synthetic batches already resident in HBM (the metric is
defined over the wall time of the client's train()); one "step" = one
forward_backward of that epoch: CLIP ViT-B/16 image encoder with FairLoRA r=8 (G=3) + text
tower + logits head + cross-entropy, full backward (dX + LoRA/ctx gradients)
and the fused SGD-momentum update, bf16 activations/frozen weights with fp32
accumulation and fp32 trainable tensors, batch 32 of 224x224x3
(BASELINE.json configs[1]).  With N > 1 every rank is one federated client
(one process per GPU, weak scaling) and the K steps end with the round-boundary
FedAvg all-reduce of the LoRA/ctx parameters over RCCL.  `python bench.py --gpus N`
starts the N ranks itself (torch.distributed.run children); started under
torch.distributed.run it is one of the ranks and takes the world size from the
launcher (an explicit --gpus that disagrees is an error).

    --config c2   (default) BASELINE.json configs[1]: the headline workload above
    --config c4   configs[3]: 3D OCT, 4 volumes of 200x224x224 per step (D=8 -> 100
                  ViT-B/16 images), FairLoRA r=16, dX through the trainable slice conv
    --config c5   configs[4]: RN50 FairLoRA r=8, gender (2 groups), bs 32; the round
                  boundary also averages the BatchNorm buffers (second all-reduce)
`value` of c4 / c5 is in that workload's own unit (volumes/sec, images/sec); the
driver's line is c2.  The default c2 run reports c4 and c5 under `secondary`
(child processes, each with its own `roofline`).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      achieved TFLOP/s of the dominant kernel (the MFMA GEMM), timed
                per launch with HIP events in a second pass over the same K steps
  cpu_baseline  the oracle (CPU restatement of the reference) timed on the host
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The engine runs three HIP streams (vision chain, text tower, LoRA-gradient reductions) and RCCL adds its
# own; with ROCm's default of 4 hardware queues they get multiplexed and the overlap is lost (measured:
# 9.4 vs 6.9 ms/step once a communicator exists).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import socket
import subprocess

import torch
import torch.distributed as dist

BATCH = 32
MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU); default: the launcher's WORLD_SIZE, else 1")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"],
                    help="BASELINE.json workload: c2 = configs[1] ViT-B/16 r=8 bs32 (the metric), c4 = configs[3] 3D OCT "
                         "r=16, c5 = configs[4] RN50 r=8 G=2")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="storage type: bf16 (BASELINE.json configs[1]), f16 (IEEE half, the reference's PREC=fp16), f32")
    ap.add_argument("--rank", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trainer", action="store_true", help="skip the GLP_OT_SVLoRA.train() throughput entries")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[3] (3D OCT) / configs[4] (RN50) step times (child processes after the timed region)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--engine-step", action="store_true",
                    help="configs[1]: time the bare engine loop (forward_backward + sgd_step) instead of the trainer's per-batch path")
    ap.add_argument("--serial", action="store_true",
                    help="fold the side streams into the main stream for the timed steps (one kernel at a time): the "
                         "mode the roofline pass measures in; use it under rocprofv3 to get per-kernel durations "
                         "that are comparable with roofline.avg_launch_us")
    ap.add_argument("--launch", default="replay", choices=["replay", "eager", "graph"],
                    help="replay: recorded launch plan (default); eager: Python per launch; graph: one hipGraph per step")
    return ap.parse_args()


class GemmTimer:
    """Wraps ops.gemm_nt (and ops.conv3x3, RN50's implicit-GEMM convolutions): one HIP event pair per launch on the
    launching stream."""

    def __init__(self, ops):
        self.ops, self.orig, self.orig_conv, self.rec, self.empty = ops, ops.gemm_nt, ops.conv3x3, [], []

    def _pair(self):
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        def timed(a, b, out, **kw):
            e0, e1 = self._pair()
            e0.record()
            r = self.orig(a, b, out, **kw)
            e1.record()
            M, K = a.shape
            N = b.shape[0]
            fl = 2.0 * M * N * K
            rk = kw.get("rankop")
            if kw.get("ts") is not None:
                fl += 2.0 * M * N * kw["ts"].shape[1]
            if rk is not None:                       # in-GEMM down projection (16 padded rank rows) + rank-r update
                fl += 2.0 * M * K * 16 + 2.0 * M * N * rk.S.shape[1]
                if getattr(rk, "lgrad", None) is not None:      # FFM_EPI_LGRAD: dB(c_fc) and dA(c_proj), 2 M N r each
                    fl += 2 * 2.0 * M * N * rk.S.shape[1]
            # algorithmic HBM bytes of the launch: each operand once, the stored output, and every full-size epilogue
            # stream (residual in, GELU image out, pre-activation in for dGELU) - DESIGN.md section 4.1's count
            es = a.element_size()
            streams = 1 + sum(kw.get(k) is not None for k in ("res", "gelu_out", "dgelu_aux"))
            self.rec.append((e0, e1, fl, M, es * (M * K + N * K + streams * M * N)))
            return r

        def timed_conv(x, w, out, *a, **kw):
            e0, e1 = self._pair()
            e0.record()
            r = self.orig_conv(x, w, out, *a, **kw)
            e1.record()
            self.rec.append((e0, e1, 2.0 * x.shape[0] * w.shape[0] * 9 * x.shape[1], x.shape[0],
                             x.element_size() * (x.numel() + w.numel() + x.shape[0] * w.shape[0])))
            return r
        self.ops.gemm_nt, self.ops.conv3x3 = timed, timed_conv
        return self

    def __exit__(self, *a):
        self.ops.gemm_nt, self.ops.conv3x3 = self.orig, self.orig_conv

    def calibrate(self, k=8):
        """k event pairs with nothing between them, in the same backed-up queue: what a pair costs by itself."""
        for _ in range(k):
            e0, e1 = self._pair()
            e0.record()
            e1.record()
            self.empty.append((e0, e1))

    def pair_floor_us(self):
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) for e0, e1 in self.empty)
        return t[len(t) // 2] * 1e3 if t else None

    def summary(self, min_rows=0):
        torch.cuda.synchronize()
        sel = [(e0.elapsed_time(e1), f, by) for e0, e1, f, M, by in self.rec if M >= min_rows]
        self.alg_bytes = sum(by for _, _, by in sel) / max(1, len(sel))
        return len(sel), sum(t for t, _, _ in sel), sum(f for _, f, _ in sel)


def usable_cores() -> int:
    """Cores this process may actually run on: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes the BLAS thread pool)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(mcfg, budget_s=40.0):
    """Oracle train step (fp32, bs 32) on the host cores: 1 warm-up + 5 timed steps, median;
    stops early past `budget_s` seconds so that the default run stays within minutes on a slow host."""
    from fairfedmed_amd import synth
    from oracle import fairlora_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
    batch = synth.make_batch(mcfg, BATCH, seed=1234)
    keys = synth.trainable_keys(mcfg)
    opt = O.SgdState()
    t_start = time.time()
    O.train_step(sd, opt, batch, mcfg, keys)                     # warm-up
    times = []
    for _ in range(5):
        t0 = time.time()
        O.train_step(sd, opt, batch, mcfg, keys)
        times.append(time.time() - t0)
        if time.time() - t_start > budget_s:
            break
    med = sorted(times)[len(times) // 2]
    return {"value": BATCH / med, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(),
            "sample": f"{len(times)} timed steps (after 1 warm-up) of the oracle's fp32 train step, batch {BATCH}, "
                      f"same ViT-B/16 FairLoRA r=8 workload; median {med:.3f} s/step"}


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run and
    relay their output and exit code.  Nothing in THIS process has touched the GPU (importing torch does not
    initialise HIP; counting the devices may, on builds that fall back to hipGetDeviceCount), and whatever it has
    touched the ranks are always FRESH subprocess children, never an exec of this process."""
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("FFM_BENCH_ONE_DEVICE"):
        raise SystemExit(f"--gpus {n}: only {have} GPU(s) visible on this node")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    for line in proc.stdout:                         # rank 0's JSON line (and anything else the ranks print)
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


TRAIN_STEPS = 32          # steps per round = 32 (1024 images per client-round)


def build_bench_trainer(mcfg, sd, args, dev, rank):
    """GLP_OT_SVLoRA through the registry with the reference's config field names: ViT-B/16, FairLoRA, race (3 groups),
    one client with 32 resident batches of 32 (one client-round of the metric)."""
    from types import SimpleNamespace as NS
    from fairfedmed_amd import config as C
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData
    import fairfedmed_amd.trainer  # noqa: F401  (registers GLP_OT_SVLoRA)
    data = SyntheticFedData(mcfg, 1, train_batches=max(TRAIN_STEPS if not args.no_trainer else 1, args.steps, args.warmup, 1),
                            test_batches=1, batch_size=BATCH, seed=1234 + 7919 * rank, device=dev)
    # Two more "clients" over the same resident batches: an epoch of exactly W batches (warm-up) and one of exactly K (the
    # timed GLP_OT_SVLoRA.train() call of the headline); client 0 keeps the 32-batch round of SURVEY.md section 8(d).
    from fairfedmed_amd.trainer import _ListDataset, _Loader
    ds0 = data.fed_train_loader_x_dict[0].dataset
    for name, n in (("bench_warmup", args.warmup), ("bench_timed", args.steps)):
        data.fed_train_loader_x_dict[name] = _Loader(_ListDataset(ds0.batches[:n], ds0.attributes, ds0.num_groups))
    if not args.no_trainer:
        data.fed_train_loader_x_dict[0] = _Loader(_ListDataset(ds0.batches[:TRAIN_STEPS], ds0.attributes, ds0.num_groups))
    cfg = NS(SEED=1, OUTPUT_DIR="", DEVICE=dev, VERBOSE=False,
             INPUT=NS(SIZE=(224, 224), PIXEL_MEAN=list(C.CLIP_PIXEL_MEAN), PIXEL_STD=list(C.CLIP_PIXEL_STD)),
             DATASET=NS(NAME="FairFedMed", ATTRIBUTES=["race"], ATTRIBUTE_TYPE="race", MODALITY_TYPE="slo_fundus",
                        DIM_PER_3D_SLICE=0, USERS=1),
             MODEL=NS(BACKBONE=NS(NAME="ViT-B/16"), STATE_DICT=sd),
             TRAINER=NS(NAME="GLP_OT_SVLoRA", LAMBDA_FAIRNESS=0.0,
                        GLP_OT=NS(N=2, N_CTX=4, PREC={"bf16": "bf16", "f16": "fp16", "f32": "fp32"}[args.dtype], OT="None"),
                        GLP_OT_LORA=NS(RANK=args.rank, ALPHA=2.0, TYPE="FairLoRA", GLOBAL_S=False, DISABLE_ATTR=False,
                                       UNFREEZE_IMAGE_ENCODER=True)),
             OPTIM=NS(NAME="sgd", LR=1e-3, MOMENTUM=0.9, WEIGHT_DECAY=5e-4, LR_SCHEDULER="single_step", STEPSIZE=200,
                      GAMMA=0.1, MAX_EPOCH=1),
             DATALOADER=NS(TRAIN_X=NS(BATCH_SIZE=BATCH)), TEST=NS(BATCH_SIZE=BATCH, NO_TEST=True),
             TRAIN=NS(METRICS_EVERY=1, CHECKPOINT_FREQ=0), DATA=data)
    return build_trainer(cfg)


def trainer_throughput(tr, use_dist):
    """images/sec of GLP_OT_SVLoRA.train(idx=0): one local epoch of TRAIN_STEPS batches through run_epoch /
    forward_backward / model_update / update_lr (TrainerX.run_epoch, Dassl/dassl/engine/trainer.py:685-741; the function
    SURVEY.md section 8(d) defines the metric over).  Three settings of the per-step reporting:
      default       the reference's summary dict every step, its values left on the GPU until read (one sync per epoch)
      every_32      cfg.TRAIN.METRICS_EVERY = 32
      sync_per_step the reference's literal behaviour: host sync + host-side (sklearn-equivalent) AUC every step"""
    res = {}
    modes = (("default", dict(METRICS_EVERY=1, SYNC_EVERY_STEP=False, HOST_METRICS=False)),
             ("every_32", dict(METRICS_EVERY=TRAIN_STEPS, SYNC_EVERY_STEP=False, HOST_METRICS=False)),
             ("sync_per_step", dict(METRICS_EVERY=1, SYNC_EVERY_STEP=True, HOST_METRICS=True)))
    for name, knobs in modes:
        for k, v in knobs.items():
            setattr(tr.cfg.TRAIN, k, v)
        tr.train(idx=0, global_epoch=0, is_fed=True)                  # warm-up epoch (records the launch plan)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter()
        tr.train(idx=0, global_epoch=0, is_fed=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=tr.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        res[name] = {"images_per_sec_per_client": BATCH * TRAIN_STEPS / dt, "ms_per_step": dt / TRAIN_STEPS * 1e3}
    for k, v in modes[0][1].items():
        setattr(tr.cfg.TRAIN, k, v)
    res["steps_per_round"] = TRAIN_STEPS
    res["what"] = "wall time of GLP_OT_SVLoRA.train(idx) for one local epoch of 32 resident batches of 32, after one warm-up epoch"
    return res


def secondary_configs():
    """BASELINE.json configs[3] (3D OCT r=16) and configs[4] (RN50 r=8 G=2) through `bench.py --config c4 / c5`, each in a
    child process of its own (fresh HIP queues; this process is done timing), each with its own `roofline`.  Never part
    of `value`."""
    out = {}
    # configs[4] is quoted in fp16 since round 6: BASELINE.json names no precision for it, fp16 is the reference's own PREC
    # (federated_main.py:85), it runs the same kernels at the same MFMA rate as bf16 - and it is the 16-bit mode in which the RN
    # tower meets north_star's AUC +-0.002 against the fp32-weights reference (tests/test_auc_parity_gpu.py; the bf16 RN MODEL,
    # frozen weights rounded to 8 significant bits, already sits 0.002 from it before any arithmetic)
    for key, cfg, dt in (("configs[3]_oct3d_vitb16_r16_bf16", "c4", "bf16"), ("configs[4]_rn50_r8_g2_fp16", "c5", "f16")):
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", cfg, "--gpus", "1", "--steps", "10",
                                "--warmup", "3", "--no-cpu-baseline", "--no-secondary", "--no-trainer", "--dtype", dt],
                               capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if line:
                j = json.loads(line[-1])
                out[key] = {"workload": j["config"]["workload"], "ms_per_step": j["ms_per_step"], "value": j["value"],
                            "unit": j["unit"], "steps": j["steps"], "dtype": j["dtype"],
                            "trainable_elems": j["config"]["trainable_elems"], "final_loss": j["config"]["final_loss"],
                            "roofline": j.get("roofline")}
                for extra in ("vit_images_per_sec",):
                    if extra in j["config"]:
                        out[key][extra] = j["config"][extra]
            else:
                out[key] = {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:                                        # a failed side measurement never fails the bench
            out[key] = {"error": repr(e)[:300]}
    return out


class Workload:
    """One BASELINE.json configuration: engine, resident batch, and how a step's work is counted."""

    def __init__(self, args, dev, rank):
        import dataclasses
        from fairfedmed_amd import config as C
        from fairfedmed_amd import synth
        from fairfedmed_amd.engine import FairLoRAEngine
        dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
        self.key, self.tr, self.vit_images = args.config, None, None
        if args.config == "c2":
            self.mcfg = C.vit_b16(rank=args.rank)
            sd = synth.make_state_dict(self.mcfg, seed=1, lora_init="reference")
            # The engine is the one the registry-built trainer owns (build_trainer(cfg) -> GLP_OT_SVLoRA.build_model), so
            # the headline steps and the trainer-level entries run on the SAME engine, streams and hardware queues.
            self.tr = build_bench_trainer(self.mcfg, sd, args, dev, rank)
            self.eng = self.tr.engine
            assert isinstance(self.eng, FairLoRAEngine)
            batch = synth.make_batch(self.mcfg, BATCH, seed=1234 + rank)
            self.units, self.unit = BATCH, "images/sec"
            self.metric = "images/sec per client-round, ViT-B/16 FairLoRA r=8, bs=32 224^2"
            self.desc = ("configs[1]: 1-client ViT-B/16 FairLoRA rank=%d G=3, bs=32 synthetic 224x224x3, fwd+bwd+SGD per step"
                         % args.rank)
            self.min_rows, self.rows = 1024, BATCH * 197
            self.by_attr = [400, 300, 324]
        elif args.config == "c4":
            B, S = 4, 25
            self.mcfg = dataclasses.replace(C.vit_b16(rank=16), dim_per_3d_slice=8)
            sd = synth.make_state_dict(self.mcfg, seed=1, lora_init="reference")
            self.eng = FairLoRAEngine(self.mcfg, sd, dtype=dtype, max_images=B * S, device=dev)
            batch = synth.make_batch(self.mcfg, B, seed=3 + rank, slices=S, signal=0.2)
            self.units, self.unit, self.vit_images = B, "volumes/sec", B * S
            self.metric = "volumes/sec per client-round, 3D OCT 200x224x224, ViT-B/16 FairLoRA r=16"
            self.desc = ("configs[3]: 3D OCT, 4 volumes of 200x224x224 per step, D=8 -> 100 ViT-B/16 images, FairLoRA r=16 G=3, "
                         "fwd+bwd (through the trainable slice conv)+SGD per step")
            self.min_rows, self.rows = 1024, B * S * 197
            self.by_attr = [400, 300, 324]
        else:
            from fairfedmed_amd.engine_rn import create_engine
            self.mcfg = C.rn50(rank=8, num_groups=2)
            sd = synth.make_state_dict(self.mcfg, seed=1, lora_init="random")
            self.eng = create_engine(self.mcfg, sd, dtype=dtype, max_images=BATCH, device=dev)
            batch = synth.make_batch(self.mcfg, BATCH, seed=1234 + rank)
            self.units, self.unit = BATCH, "images/sec"
            self.metric = "images/sec per client-round, RN50 FairLoRA r=8 G=2, bs=32 224^2"
            self.desc = "configs[4]: RN50 (3,4,6,3) FairLoRA r=8, gender (2 groups), bs=32 synthetic 224x224x3, fwd+bwd+SGD per step"
            self.min_rows, self.rows = 1024, None
            self.by_attr = [600, 424]
        self.dtype = dtype
        self.img = batch["img"].to(dev)
        self.attr = batch["attrs"].t()[0].contiguous().to(dev)
        self.attrs = batch["attrs"].to(dev)                    # [B, n_attr]: the trainer's batch format
        self.label = batch["label"].to(dev)
        self.has_buf = hasattr(self.eng, "buffers_flat")       # RN50: BatchNorm running statistics


def main():
    args = parse()
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # started by torch.distributed.run
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if args.gpus is None:
        args.gpus = world                                # under a launcher: its world size; else one GPU
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not launched and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    use_dist = (world > 1 or launched) and not os.environ.get("FFM_BENCH_NO_DIST")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # FFM_BENCH_ONE_DEVICE=1 (test rig only): all ranks share cuda:0 and talk over gloo, so the N > 1 code path
        # can be exercised on a one-GPU box; the number it prints is meaningless
        one_dev = bool(os.environ.get("FFM_BENCH_ONE_DEVICE"))
        if one_dev:
            local = 0
        torch.cuda.set_device(local)
        if one_dev:
            dist.init_process_group("gloo")
        elif os.environ.get("FFM_BENCH_LAZY_PG"):
            dist.init_process_group("nccl")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)

    from fairfedmed_amd import config as C
    from fairfedmed_amd import ops
    from fairfedmed_amd.fedavg import FedAvgAggregator

    wl = Workload(args, dev, rank)
    mcfg, eng, tr, dtype = wl.mcfg, wl.eng, wl.tr, wl.dtype
    img, attr, label = wl.img, wl.attr, wl.label
    opt = C.OptimCfg()
    agg = None
    if use_dist:
        # the round boundary the product's rank driver runs (federated.run_fedotplora_ranks): the SAME aggregator objects
        agg = FedAvgAggregator(eng.params.flat, eng.params.offsets, mcfg.lora.num_groups, mcfg.lora.rank)
        agg_buf = (FedAvgAggregator(eng.buffers_flat(), {}, mcfg.lora.num_groups, mcfg.lora.rank, shared_half_s=False)
                   if wl.has_buf else None)
    n_client = [1024] * world
    by_attr = [wl.by_attr] * world
    buf_bytes = 0

    def eager_step():
        eng.forward_backward(img, attr, label)
        # model_update as the reference runs it: its one optimizer is registered under two names and stepped for each
        # (trainers/GLP_OT_SVLoRA.py:866-870, Dassl/dassl/engine/trainer.py:333-337); one fused launch here
        eng.sgd_step(opt.lr, opt.momentum, opt.weight_decay, repeats=2)

    eng.use_replay = args.launch == "replay"
    if args.serial:
        eng.set_overlap(False)
    if args.launch == "graph" and args.config != "c2":
        raise SystemExit("--launch graph: the captured step exists for --config c2 only")
    graphed = eng.capture_train_step(BATCH, opt.lr, opt.momentum, opt.weight_decay) if args.launch == "graph" else None

    # SURVEY.md section 8(d) defines the metric on the wall time of the client's train(): for configs[1] the timed region
    # IS one GLP_OT_SVLoRA.train(idx) call - TrainerX.run_epoch (Dassl/dassl/engine/trainer.py:685-741) over a local epoch
    # of exactly K batches resident in HBM: parse_batch_train, forward_backward (engine step + the reference's double
    # optimizer step), the per-step summary, the StepLR move and the finite check at the end of the epoch.  The W warm-up
    # steps are a train() call over W batches.  The bare engine loop of earlier rounds is reported beside it as
    # `engine_only`.
    trainer_step = tr is not None and graphed is None and not args.engine_step

    def step():
        if graphed is None:
            eager_step()
        else:
            graphed.run(img, attr, label)          # copies the (resident) batch into the graph's inputs, replays

    def run_steps(n, which):
        if trainer_step:
            if n > 0:
                tr.train(idx=which, global_epoch=0, is_fed=True)
        else:
            for _ in range(n):
                step()

    def round_boundary():
        nonlocal buf_bytes
        if agg is None:
            return
        agg.aggregate(rank, list(range(world)), n_client, by_attr, 1, 50)
        if wl.has_buf:
            # RN50: the BatchNorm running statistics / counters are state_dict entries too and are averaged with the plain
            # n_k / sum n weights (utils/fed_utils.py:76-86): a second, small all-reduce
            agg_buf.begin()
            agg_buf.add(eng.buffers_flat(), rank, list(range(world)), n_client, None)
            buf = agg_buf.finish(1, 50, grouped=False)
            eng.load_buffers_flat(buf)
            buf_bytes = buf.numel() * 4

    run_steps(args.warmup, "bench_warmup")
    round_boundary()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, "bench_timed")
    round_boundary()
    t_enqueue = time.perf_counter() - t0            # host time to enqueue the K steps (no sync inside)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    finite = int(eng.finite)
    loss = float(eng.loss)
    engine_only = None
    if trainer_step:
        # the bare engine loop on the same engine (earlier rounds' headline): K steps, same bracket, max over ranks
        tr.check_finite()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eager_step()
        torch.cuda.synchronize()
        de = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([de], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            de = float(t)
        engine_only = {"value": wl.units * args.steps * world / de, "ms_per_step": de / args.steps * 1e3,
                       "what": "eng.forward_backward + eng.sgd_step(repeats=2) in a bare loop: no parse_batch, no per-step summary"}
    # round-boundary exchange on its own: pre-scale, ONE all-reduce of the flat trainable buffer,
    # shared_half_s + EMA (RN50: + the BatchNorm-buffer all-reduce); median of 5 after the timed region, max over ranks
    fedavg_us = None
    if agg is not None:
        samples = []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            round_boundary()
            torch.cuda.synchronize()
            samples.append(time.perf_counter() - t1)
        t = torch.tensor([sorted(samples)[2]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        fedavg_us = float(t) * 1e6

    roof = None
    if not args.no_roofline and rank == 0:
        eng.use_replay = False                         # per-launch events need eager launches ...
        eng.set_overlap(False)                         # ... and one kernel at a time (no side streams)
        eager_step()
        # The event pairs must measure the GPU, not the host: with an idle queue the first event of a pair executes the
        # moment it arrives and the pair's interval then includes the Python time between its enqueue and the launch's
        # (on a slow host the figure dropped from 0.23 to 0.19 of peak with identical kernels).  A spin kernel in front of
        # every timed step keeps the queue backed up while the host enqueues the step, so the pairs run back to back.
        spin = int(2.0e9 * (0.015 if args.config == "c2" else 0.030))   # longer than one eager step's enqueue
        with GemmTimer(ops) as gt:
            for _ in range(args.steps):
                torch.cuda._sleep(spin)
                eager_step()
                gt.calibrate()
            n_all, ms_all, fl_all = gt.summary()
            n, ms, fl = gt.summary(min_rows=wl.min_rows)   # the vision tower's GEMMs (not the text tower's 40 rows)
            alg_bytes = gt.alg_bytes
            floor_us = gt.pair_floor_us()
        eng.set_overlap(True)
        peak = MFMA_BF16_PEAK_TFLOPS if dtype != torch.float32 else MFMA_F32_PEAK_TFLOPS     # (the f16 MFMA forms run at the bf16 rate)
        ach = fl / (ms * 1e-3) / 1e12
        # HBM-side bytes per launch of the same kernels come from the committed PMC passes (rocprofv3 cannot be
        # driven from inside the timed process): tools/pmc_traffic.py -> profiles/rNN_traffic.json
        traffic, tsrc = None, None
        try:
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            # configs[1]: profiles/rNN_traffic.json; configs[3] / [4]: profiles/rNN_traffic_c4.json / _c5.json (tools/pmc.sh <name> --config c4)
            suffix = "_traffic.json" if args.config == "c2" else f"_traffic_{args.config}.json"
            cand = sorted(f for f in os.listdir(pdir) if f.endswith(suffix))
            if cand and dtype != torch.float32 and (args.rank == 8 or args.config != "c2"):
                traffic = json.load(open(os.path.join(pdir, cand[-1])))["traffic_bytes_per_launch"]
                tsrc = "profiles/" + cand[-1]
        except (OSError, KeyError, ValueError):
            pass
        if args.config == "c5":
            kernel = ("ffm_gemm_nt (1x1 convolutions with the FairLoRA epilogues, attention pool) and ffm_conv3x3_nhwc "
                      "(implicit-GEMM 3x3 convolutions) of the RN50 tower: gemm_nt_kernel / conv_narrow_kernel (%s)" % args.dtype)
        else:
            kernel = ("ffm_gemm_nt on the vision tower (M=%d): gemm_panel_kernel<%s> on fragment-packed frozen "
                      "weights (bf16; the qkv / c_fc launches also carry ln_1 / ln_2, FFM_EPI_LNIN, whose flops are "
                      "not counted), gemm_nt_kernel otherwise" % (wl.rows, args.dtype))
        roof = {"bound": "mfma", "kernel": kernel,
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                "traffic_unit": "HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE), mean over the same launches",
                # what the same launches must move at least (operands once + stored output + full-size epilogue streams),
                # mean per launch, for THIS workload - the figure `traffic` is to be read against
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic_source": tsrc,
                # machine-readable provenance: the PMC passes are separate rocprofv3 runs of this same command (tools/pmc.sh),
                # committed under profiles/ - the figure is NOT collected inside this timed process
                "traffic_measured_in_run": False,
                "launches_per_step": n // args.steps, "avg_launch_us": ms * 1e3 / n,
                "gemm_ms_per_step": ms / args.steps,
                # what an event pair measures with NOTHING between the two records (median, same backed-up queue):
                # the part of avg_launch_us that is the events' own, which rocprofv3's kernel durations do not contain
                "event_pair_floor_us": floor_us,
                "frac_net_of_event_floor": (fl / ((ms * 1e-3) - n * floor_us * 1e-6) / 1e12 / peak) if floor_us else None,
                "all_gemm_launches": {"launches_per_step": n_all // args.steps, "achieved": fl_all / (ms_all * 1e-3) / 1e12,
                                      "gemm_ms_per_step": ms_all / args.steps,
                                      "note": "includes the text tower's 96 latency-bound launches on 40 rows (4 prompts x 10 tokens)"},
                # the whole step against the same peak: SURVEY section 8(d)'s 72.0 GFLOP per image and train step (vision
                # forward + LoRA-only backward + text tower) x images per step / the timed step - every kernel, boundary and
                # side stream included (configs[1] only: the figure is that workload's)
                "whole_step": ({"gflop_per_image": 72.0, "achieved": 72.0e9 * wl.units / (dt / args.steps) / 1e12,
                                "frac": 72.0e9 * wl.units / (dt / args.steps) / 1e12 / peak} if args.config == "c2" and args.rank == 8 else None),
                # since round 6 the dX launches of c_proj / c_fc / qkv also carry the two LayerNorm backwards of a block
                # (FFM_EPI_LNB_STAT / FFM_EPI_LNB_APPLY: 22 layernorm_bwd launches of 9.9 us folded into 33 of these 96
                # launches); their flops are not counted, so `frac` prices the launches' whole duration against the GEMM
                # flops alone - it moved 0.241 -> 0.230 while the step got 2.4 % shorter
                "note": "launch durations include the LayerNorm-backward work folded into dX(c_proj) / dX(c_fc) / dX(qkv) (uncounted flops)",
                "measured": "HIP events around every ffm_gemm_nt launch, second pass over the same K steps with the "
                            "side streams folded into the main stream (one kernel at a time); value comes from the "
                            "un-instrumented overlapped pass"}
    trainer_res = None
    if not args.no_trainer and tr is not None:
        eng.set_overlap(True)
        eng.use_replay = True
        trainer_res = trainer_throughput(tr, use_dist)
    if use_dist:
        dist.barrier()

    if rank == 0:
        res = {
            "metric": wl.metric,
            "value": wl.units * args.steps * world / dt,
            "unit": wl.unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": wl.desc + ("" if world == 1 else "; one client per GPU, FedAvg all-reduce at the round end"),
                       "global_batch": wl.units * world, "clients": world,
                       "trainable_elems": eng.params.numel, "final_loss": loss, "loss_finite": finite,
                       # (train() ends with the epoch's one host sync, so in the trainer mode this is the epoch's wall time, not the enqueue time)
                       "host_enqueue_ms_per_step": None if trainer_step else t_enqueue / args.steps * 1e3,
                       "launch": args.launch,
                       "timed_step": ("one GLP_OT_SVLoRA.train(idx) call: run_epoch over a local epoch of K resident batches "
                                      "(SURVEY.md section 8(d)); ms_per_step = its wall time / K"
                                      if trainer_step else "engine.forward_backward + sgd_step"),
                       "rccl_ranks": dist.get_world_size() if use_dist else 1,
                       "backend": dist.get_backend() if use_dist else None},
        }
        if wl.vit_images:
            res["config"]["vit_images_per_sec"] = wl.vit_images * args.steps * world / dt
        if fedavg_us is not None:
            res["config"]["fedavg_round_boundary_us"] = fedavg_us
            res["config"]["fedavg_payload_bytes"] = eng.params.numel * 4
            if wl.has_buf:
                res["config"]["fedavg_buffer_payload_bytes"] = buf_bytes
        if engine_only:
            res["engine_only"] = engine_only
        if roof:
            res["roofline"] = roof
        if trainer_res:
            res["trainer"] = trainer_res
        if world == 1 and not args.no_secondary and args.config == "c2":
            res["secondary"] = secondary_configs()
        if world == 1 and not args.no_cpu_baseline and args.config == "c2":
            res["cpu_baseline"] = cpu_baseline(mcfg)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
