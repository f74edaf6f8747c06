"""``GLP_OT_SVLoRA`` trainer: the reference's trainer API (the methods
``federated_main.py`` calls) driving the HIP engine.

Reference: trainers/GLP_OT_SVLoRA.py:767-1053 (trainer), Dassl/dassl/engine/
trainer.py:108-342,345-589,682-741 (TrainerBase / SimpleTrainer / TrainerX).

``cfg`` is any attribute tree with the reference's yacs field names (a yacs
CfgNode or types.SimpleNamespace); only the fields the hot path reads are used.
Data loading is out of scope: the trainer takes
per-client loaders of batch dicts {"img","label","attrs"} through ``data``
(see ``SyntheticFedData``).
"""
from __future__ import annotations

import os
import time
from collections import OrderedDict
from types import SimpleNamespace as NS
from collections.abc import Mapping
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import config as C
from . import synth
from . import ops
from .metrics import (auc_macro_ovr, basic_from_counts, comprehensive_scores, comprehensive_scores_from_counts,
                      macro_f1)
from .model import CustomCLIP
from .registry import TRAINER_REGISTRY

ATTRIBUTE_GROUPS = {   # trainers/GLP_OT_SVLoRA.py:775-790
    "FairFedMed": {"race": ["Asian", "Black", "White"], "language": ["English", "Spanish", "Others"],
                   "ethnicity": ["Non-hispanic", "Hispanic"], "gender": ["Male", "Female"]},
    "FedChexMimic": {"race": ["White", "Asian", "Black"], "gender": ["Male", "Female"], "age": ["0-60", "60+"]},
}

# EOT positions of the tokenised prompts "X X X X <classname>." (SURVEY.md §8(c) (iii)); the BPE
# tokenizer itself is init-time only and out of scope.
KNOWN_EOT = {"NOT Glaucoma": 9, "Glaucoma": 8, "NOT Pleural Effusion": 11, "Pleural Effusion": 10}


class _ListDataset:
    def __init__(self, batches: List[dict], attributes: Sequence[str], num_groups: Dict[str, int]):
        self.batches, self.attributes, self.num_groups = batches, list(attributes), num_groups

    def __len__(self):
        return sum(int(b["label"].shape[0]) for b in self.batches)

    def count_by_attribute(self, attr: str) -> List[int]:
        """Samples per demographic group (Dassl/dassl/data/data_manager.py:435-473)."""
        col = self.attributes.index(attr)
        cnt = np.zeros(self.num_groups[attr], dtype=np.int64)
        for b in self.batches:
            v, c = np.unique(b["attrs"][:, col].cpu().numpy(), return_counts=True)
            cnt[v] += c
        return cnt.tolist()


class _Loader:
    def __init__(self, dataset: _ListDataset):
        self.dataset = dataset

    def __iter__(self):
        return iter(self.dataset.batches)

    def __len__(self):
        return len(self.dataset.batches)


MODALITIES_3D = {"oct_bscans", "oct_bscans_3d", "mac_onh", "onh_mac"}


class _DeviceSummary(Mapping):
    """forward_backward's {"loss", "acc", "auc"} (trainers/GLP_OT_SVLoRA.py:959-970) with the values still on the GPU:
    the loss, the batch's integer evaluator counts (ffm_eval_counts: accuracy and the exact rank AUC are ratios of
    them) and the sticky finite flag.  Reading any entry copies ~100 bytes to the host once and yields the same floats
    the reference computes with loss.item(), compute_accuracy and sklearn's roc_auc_score; until then the step costs
    no host synchronisation (the reference pays three per step)."""

    def __init__(self, trainer, loss: torch.Tensor, counts: torch.Tensor):
        self._tr, self._loss, self._counts, self._vals = trainer, loss, counts, None

    def _materialise(self) -> dict:
        if self._vals is None:
            self._tr.check_finite()
            row = self._counts[-1].cpu().numpy()                         # the 'all' row
            res = basic_from_counts(row[None])
            self._vals = {"loss": float(self._loss), "acc": res[0], "auc": res[3] / 100.0}
            self._loss = self._counts = None
        return self._vals

    def __getitem__(self, k):
        return self._materialise()[k]

    def __iter__(self):
        return iter(("loss", "acc", "auc"))

    def __len__(self):
        return 3

    def __repr__(self):
        return repr(self._materialise())


class SyntheticFedData:
    """Per-client train/test loaders of synthetic batches (what the reference's DataManager
    exposes as fed_{train,test}_loader_x_dict, Dassl/dassl/data/data_manager.py:104-133)."""

    def __init__(self, mcfg: C.ModelCfg, num_clients: int, train_batches: int, test_batches: int, batch_size: int,
                 attribute: str = "race", classnames=("NOT Glaucoma", "Glaucoma"), seed: int = 1234,
                 signal: float = 0.25, device: str = "cpu", test_batch_size: Optional[int] = None, overlap: float = 0.0):
        ng = {attribute: mcfg.lora.num_groups}
        mk = lambda s, n=batch_size: {k: v.to(device) for k, v in synth.make_batch(mcfg, n, seed=s, signal=signal, overlap=overlap).items()}
        tbs = test_batch_size or batch_size
        self.fed_train_loader_x_dict, self.fed_test_loader_x_dict = {}, {}
        for c in range(num_clients):
            tr = [mk(seed + 1000 * c + i) for i in range(train_batches)]
            te = [mk(seed + 1000 * c + 500 + i, tbs) for i in range(test_batches)]
            self.fed_train_loader_x_dict[c] = _Loader(_ListDataset(tr, [attribute], ng))
            self.fed_test_loader_x_dict[c] = _Loader(_ListDataset(te, [attribute], ng))
        self.dataset = NS(classnames=list(classnames))
        self.num_classes = len(classnames)
        self.lab2cname = {i: n for i, n in enumerate(classnames)}
        self.classnames = list(classnames)


def resolve_precision(prec: str):
    """TRAINER.GLP_OT.PREC -> storage dtype of the engine.  The reference knows 'fp16' (its GPU default,
    federated_main.py:85: convert_weights halves the CLIP weights, clip/model.py:609-630), 'fp32' and 'amp'
    (fp32 weights under autocast, trainers/GLP_OT_SVLoRA.py:833-835, 900-907).  'fp16' is IEEE half here too (activations
    and frozen weights in float16, fp32 accumulation and fp32 trainable tensors, the device-side finite flag as the
    overflow guard: FFM_F16, the half twins of the 16-bit kernels); 'bf16' is MI355X's throughput mode that
    BASELINE.json configs[1] names (same kernels, 8 significand bits instead of 11, no overflow at 65504)."""
    if prec in ("fp32", "amp"):
        return torch.float32
    if prec == "bf16":
        return torch.bfloat16
    if prec == "fp16":
        return torch.float16
    raise ValueError(f"TRAINER.GLP_OT.PREC={prec!r}: expected 'fp16', 'bf16', 'fp32' or 'amp'")


@TRAINER_REGISTRY.register()
class GLP_OT_SVLoRA:
    """FairLoRA trainer.  Construct with ``GLP_OT_SVLoRA(cfg, data=..., state_dict=...)`` or through
    ``build_trainer(cfg)`` with ``cfg.DATA`` / ``cfg.MODEL.STATE_DICT`` set."""

    def __init__(self, cfg, data=None, state_dict=None):
        self._models, self._optims, self._scheds = OrderedDict(), OrderedDict(), OrderedDict()
        self.check_cfg(cfg)
        if not torch.cuda.is_available():
            raise RuntimeError("GLP_OT_SVLoRA runs on an MI355X through the HIP engine; no GPU is visible")
        self.device = torch.device(getattr(cfg, "DEVICE", "cuda:0"))
        self.cfg = cfg
        self.start_epoch = self.epoch = 0
        self.max_epoch = cfg.OPTIM.MAX_EPOCH
        self.output_dir = getattr(cfg, "OUTPUT_DIR", "")
        self.dm = data if data is not None else getattr(cfg, "DATA", None)
        if self.dm is None:
            raise ValueError("no data: pass data=SyntheticFedData(...) (dataset I/O is out of scope)")
        self.fed_train_loader_x_dict = self.dm.fed_train_loader_x_dict
        self.fed_test_loader_x_dict = self.dm.fed_test_loader_x_dict
        self.num_classes, self.lab2cname, self.classnames = self.dm.num_classes, self.dm.lab2cname, self.dm.classnames
        self._state_dict = state_dict if state_dict is not None else getattr(cfg.MODEL, "STATE_DICT", None)
        self.build_model()
        self.best_result = -np.inf
        self.time_start = self.total_time_start = time.time()

    # ------------------------------------------------------------ config --
    def check_cfg(self, cfg):
        assert cfg.TRAINER.GLP_OT.PREC in ["fp16", "fp32", "amp", "bf16"]

    def retrieval_attributes(self, attr_name: str):
        try:
            return ATTRIBUTE_GROUPS[self.cfg.DATASET.NAME][attr_name]
        except KeyError:
            raise NotImplementedError(self.cfg.DATASET.NAME)

    def model_cfg(self) -> C.ModelCfg:
        cfg = self.cfg
        lora = cfg.TRAINER.GLP_OT_LORA
        if lora.TYPE not in ("FairLoRA", "SVLoRA", "LoRA"):
            raise NotImplementedError(lora.TYPE)                       # trainers/GLP_OT_SVLoRA.py:533-534
        disable = getattr(lora, "DISABLE_ATTR", False)
        # LoRA / SVLoRA carry no per-group singular values: one group, the attribute is ignored by their forward
        G = 1 if (disable or lora.TYPE != "FairLoRA") else len(self.retrieval_attributes(cfg.DATASET.ATTRIBUTE_TYPE))
        names = list(self.dm.dataset.classnames)
        got = cfg.TRAINER.GLP_OT
        if str(got.OT) not in ("None", "Sinkhorn", "COT"):
            raise NotImplementedError(got.OT)                       # trainers/GLP_OT_SVLoRA.py:729-730
        try:
            eot = tuple(KNOWN_EOT[n.replace("_", " ")] for n in names)
        except KeyError as e:
            raise NotImplementedError(f"EOT position of class prompt {e} is not pinned (tokenizer is out of scope)")
        name = cfg.MODEL.BACKBONE.NAME
        base = C.vit_b16() if name in ("ViT-B/16", "vit_b16") else C.rn50() if name in ("RN50", "rn50") \
            else getattr(cfg.MODEL, "GEOMETRY")
        # 3D modalities go through the trainable per-slice conv (trainers/GLP_OT_SVLoRA.py:584-586)
        is_3d = getattr(cfg.DATASET, "MODALITY_TYPE", "slo_fundus") in MODALITIES_3D
        return C.ModelCfg(vision=base.vision, text=base.text,
                          lora=C.LoraCfg(rank=lora.RANK, alpha=lora.ALPHA, num_groups=G, lora_type=lora.TYPE,
                                         global_s=bool(getattr(lora, "GLOBAL_S", False))),
                          n_prompts=cfg.TRAINER.GLP_OT.N, n_ctx=cfg.TRAINER.GLP_OT.N_CTX, n_cls=len(names), eot=eot,
                          pixel_mean=tuple(cfg.INPUT.PIXEL_MEAN), pixel_std=tuple(cfg.INPUT.PIXEL_STD),
                          dim_per_3d_slice=cfg.DATASET.DIM_PER_3D_SLICE if is_3d else 0,
                          ot=str(got.OT), ot_eps=float(getattr(got, "EPS", 0.1)), ot_thresh=float(getattr(got, "THRESH", 1e-3)),
                          ot_max_iter=int(getattr(got, "MAX_ITER", 100)), ot_top_percent=float(getattr(got, "TOP_PERCENT", 1.0)))

    # ------------------------------------------------------------- model --
    def build_model(self):
        cfg = self.cfg
        mcfg = self.model_cfg()
        dtype = resolve_precision(cfg.TRAINER.GLP_OT.PREC)
        sd = self._state_dict
        if sd is None:
            # pretrained CLIP cannot be downloaded here (trainers/GLP_OT_SVLoRA.py:23-43 needs network)
            sd = synth.make_state_dict(mcfg, seed=getattr(cfg, "SEED", 1), lora_init="reference")
        bs = max(cfg.DATALOADER.TRAIN_X.BATCH_SIZE, cfg.TEST.BATCH_SIZE)
        if mcfg.dim_per_3d_slice:
            # every sample becomes C / DIM_PER_3D_SLICE ViT images; the volume depth comes from the data
            first = next(iter(self.fed_train_loader_x_dict[min(self.fed_train_loader_x_dict)]))
            bs *= first["img"].shape[1] // mcfg.dim_per_3d_slice
        self.model = CustomCLIP(mcfg, sd, dtype=dtype, max_images=bs, device=str(self.device))
        self.engine = self.model.engine
        # the per-step summary's evaluator counts ride inside the step, beside the backward pass (binary tasks, device
        # metrics: the default); TRAIN.HOST_METRICS / a fairness term keep the host path, which does not read them
        tc = getattr(cfg, "TRAIN", NS())
        if not getattr(tc, "HOST_METRICS", False) and getattr(cfg.TRAINER, "LAMBDA_FAIRNESS", 0.0) == 0.0:
            self.engine.enable_step_counts()
        self._finite_acc = torch.ones(1, device=self.device, dtype=torch.int32)
        o = cfg.OPTIM
        self.optim = NS(lr0=o.LR, momentum=o.MOMENTUM, weight_decay=o.WEIGHT_DECAY,
                        param_groups=[{"lr": o.LR}])
        stepsize = o.STEPSIZE[-1] if isinstance(o.STEPSIZE, (list, tuple)) else o.STEPSIZE
        self.sched = NS(step_size=stepsize if stepsize > 0 else o.MAX_EPOCH, gamma=o.GAMMA, last_epoch=0)
        # trainers/GLP_OT_SVLoRA.py:866-870: BOTH names are registered with the SAME optimizer and scheduler objects
        # when UNFREEZE_IMAGE_ENCODER is set (every FairLoRA script sets it)
        self.register_model("prompt_learner", self.model.prompt_learner, self.optim, self.sched)
        if getattr(cfg.TRAINER.GLP_OT_LORA, "UNFREEZE_IMAGE_ENCODER", True):
            self.register_model("image_encoder", self.model.image_encoder, self.optim, self.sched)

    def register_model(self, name="model", model=None, optim=None, sched=None):
        assert name not in self._models, "Found duplicate model names"
        self._models[name], self._optims[name], self._scheds[name] = model, optim, sched

    def get_model_names(self, names=None):
        return list(self._models.keys()) if names is None else list(names)

    def set_model_mode(self, mode="train", names=None):
        for n in self.get_model_names(names):
            self._models[n].train(mode == "train")

    def get_current_lr(self, names=None):
        return self.optim.param_groups[0]["lr"]

    def steps_per_update(self, names=None) -> int:
        """How often TrainerBase.model_update / update_lr step the ONE shared optimizer / scheduler per call: once per
        registered model name (Dassl/dassl/engine/trainer.py:253-258, 333-337), i.e. TWICE with the image encoder
        registered.  cfg.TRAINER.COMPAT_DOUBLE_STEP = False gives the single step a
        reader of the reference would expect (a deliberate divergence; the goldens pin the default)."""
        if not getattr(self.cfg.TRAINER, "COMPAT_DOUBLE_STEP", True):
            return 1
        return sum(1 for n in self.get_model_names(names) if self._optims[n] is not None)

    def update_lr(self, names=None):
        """StepLR.step() (Dassl/dassl/optim/lr_scheduler.py:100-115) at the end of a local epoch, once per registered
        name that carries the (shared) scheduler."""
        s = self.sched
        s.last_epoch += self.steps_per_update(names)
        self.optim.param_groups[0]["lr"] = self.optim.lr0 * s.gamma ** (s.last_epoch // s.step_size)

    def set_lr_epoch(self, last_epoch: int) -> None:
        """Position the (shared) StepLR as if `last_epoch` scheduler steps had been taken: the rank-parallel round loop
        calls this with the global count of client-epochs, so the schedule is the reference's whatever the world size."""
        s = self.sched
        s.last_epoch = int(last_epoch)
        self.optim.param_groups[0]["lr"] = self.optim.lr0 * s.gamma ** (s.last_epoch // s.step_size)

    def optimizer_state(self):
        """(momentum buffers fp32 [numel] on the device, [SGD steps taken, StepLR.last_epoch, lr] float64 on the same
        device): what the reference's single shared optimizer / scheduler carries from one client to the next."""
        p = self.engine.params
        scal = torch.tensor([p.steps, self.sched.last_epoch, self.optim.param_groups[0]["lr"]], dtype=torch.float64,
                            device=p.momentum.device)
        return p.momentum, scal

    def load_optimizer_state(self, momentum: torch.Tensor, scal: torch.Tensor) -> None:
        p = self.engine.params
        if momentum.data_ptr() != p.momentum.data_ptr():
            p.momentum.copy_(momentum)
        steps, last_epoch, lr = scal.cpu().tolist()
        p.steps, self.sched.last_epoch = int(steps), int(last_epoch)
        self.optim.param_groups[0]["lr"] = lr

    # ------------------------------------------------------------- batch --
    def _parse(self, batch):
        cfg = self.cfg
        image = batch["img"].to(self.device, non_blocking=True)
        label = batch["label"].to(self.device, non_blocking=True)
        attrs = batch["attrs"].to(self.device, non_blocking=True).t()
        idx = list(cfg.DATASET.ATTRIBUTES).index(cfg.DATASET.ATTRIBUTE_TYPE)
        tgt = None if getattr(cfg.TRAINER.GLP_OT_LORA, "DISABLE_ATTR", False) else attrs[idx]
        return image, label, attrs, tgt

    parse_batch_train = _parse
    parse_batch_test = _parse

    def model_inference(self, input, attr=None):
        return self.model(input, attr)

    # -------------------------------------------------------------- step --
    def forward_backward(self, batch, is_last_client=False):
        """One SGD step; returns {"loss","acc","auc"} like the reference (:959-970).  For binary tasks the returned
        mapping keeps its values on the GPU until they are read (_DeviceSummary): the reference's three host syncs per
        step (loss.item(), accuracy, sklearn AUC) shrink to one small copy per summary that is actually looked at.
        cfg.TRAIN.METRICS_EVERY = N > 1 skips the summary on the other steps altogether; TRAIN.SYNC_EVERY_STEP raises
        a non-finite loss inside the call as the reference does; TRAIN.HOST_METRICS computes the metrics with the
        host (numpy) versions."""
        image, label, _, attr = self.parse_batch_train(batch)
        # PREC 'amp': the reference's branch calls self.model(image) WITHOUT the attribute (uniform group mix) and has
        # no fairness term (trainers/GLP_OT_SVLoRA.py:890-898); autocast itself is not mirrored (fp32)
        amp = self.cfg.TRAINER.GLP_OT.PREC == "amp"
        out = self.engine.forward_backward(image, None if amp else attr, label)
        self.engine.sgd_step(self.get_current_lr(), self.optim.momentum, self.optim.weight_decay,
                             repeats=1 if amp else self.steps_per_update())     # amp: scaler.step(optim) once (:896)
        train_cfg = getattr(self.cfg, "TRAIN", NS())
        every = getattr(train_cfg, "METRICS_EVERY", 1)
        summary = {}
        want = every <= 1 or (self.batch_idx + 1) % every == 0 or (self.batch_idx + 1) == self.num_batches
        lam = getattr(self.cfg.TRAINER, "LAMBDA_FAIRNESS", 0.0)
        # sticky finite flag on the device (the reference's detect_anomaly, Dassl/dassl/engine/trainer.py:260-262):
        # raised when a summary is read, at the end of the epoch, or here with TRAIN.SYNC_EVERY_STEP
        self._finite_acc.mul_(out["finite"])
        if getattr(train_cfg, "SYNC_EVERY_STEP", False):
            self.check_finite()
        if want and out["prob"].shape[1] == 2 and (lam == 0.0 or attr is None or amp) \
                and not getattr(train_cfg, "HOST_METRICS", False):
            # (the counts come with the step when the engine was asked for them - build_model - and are copied out of its
            # static buffer like the loss; otherwise they are formed here, behind the SGD step)
            counts = out["counts"].clone() if "counts" in out else ops.eval_counts(out["prob"], label.contiguous(), None, 0)
            summary = _DeviceSummary(self, out["loss"].clone(), counts)
        elif want:
            self.check_finite()
            logits, prob = out["logits"], out["prob"]
            loss = float(out["loss"])
            if lam != 0.0 and attr is not None and not amp:             # detached fairness term (:930-948)
                correct = prob[torch.arange(len(label)), label]
                vals = torch.stack([1 - correct[attr == g].mean() for g in torch.unique(attr)])
                loss += lam * float(torch.mean(torch.abs(vals - vals.mean())))
            summary = {"loss": loss, "acc": float((logits.argmax(-1) == label).float().mean() * 100.0),
                       "auc": auc_macro_ovr(prob.cpu().numpy(), label.cpu().numpy())}
        if (self.batch_idx + 1) == self.num_batches:
            self.update_lr()
        return summary

    def run_epoch(self, idx=-1, global_epoch=-1, is_last_client=False, **_):
        self.set_model_mode("train")
        loader = self.fed_train_loader_x_dict[idx]
        self.num_batches = len(loader)
        last = {}
        for self.batch_idx, batch in enumerate(loader):
            s = self.forward_backward(batch, is_last_client=is_last_client)
            if s:
                last = s
        self.check_finite()                                           # one host sync per local epoch
        return last

    def check_finite(self) -> None:
        if int(self._finite_acc) != 1:
            self._finite_acc.fill_(1)
            # (a LOSS that is not finite, as in the reference.  fp16 GRADIENT overflow does not come here: the engine skips
            # that step and halves its device-resident gradient scale - engine.overflow_steps() counts them)
            raise FloatingPointError("Loss is infinite or NaN!")      # Dassl/dassl/engine/trainer.py:260-262

    def train(self, idx=-1, global_epoch=0, is_fed=False, is_last_client=False, **_):
        self.time_start = time.time()
        for self.epoch in range(self.start_epoch, self.max_epoch):
            self.run_epoch(idx, global_epoch, is_last_client=is_last_client)
            self.after_epoch(idx, global_epoch)
        self.after_train(idx, global_epoch, is_fed)

    def after_epoch(self, idx=-1, global_epoch=-1):
        freq = getattr(getattr(self.cfg, "TRAIN", NS()), "CHECKPOINT_FREQ", 0)
        last_epoch = (self.epoch + 1) == self.max_epoch
        if self.output_dir and ((freq > 0 and (self.epoch + 1) % freq == 0) or last_epoch):
            name = f"epoch{global_epoch}.pth" if idx == -1 else f"epoch{global_epoch}_client{idx}.pth"
            self.save_model_with_grad(os.path.join(self.output_dir, name))

    def after_train(self, idx=-1, epoch=0, is_fed=False):
        if not getattr(self.cfg.TEST, "NO_TEST", False):
            self.test(idx=idx, current_epoch=epoch)

    def save_model_with_grad(self, filename):
        """Trainable parameters + all buffers (Dassl/dassl/engine/trainer.py:177-185)."""
        sd = {n: p.detach().cpu() for n, p in self.model.named_parameters() if p.requires_grad}
        sd.update({n: b.detach().cpu() for n, b in self.model.named_buffers()})
        os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
        torch.save(sd, filename)

    def save_model(self, epoch, directory, is_best=False, model_name=""):
        """Per registered model a checkpoint `<directory>/<name>/model.pth.tar-<epoch+1>` with the sub-module's state_dict,
        the epoch and the shared optimizer / scheduler state (Dassl/dassl/engine/trainer.py:149-175,
        Dassl/dassl/utils/torchtools.py:27-80); `is_best` also writes `model-best.pth.tar`."""
        mom, scal = self.optimizer_state()
        for name in self.get_model_names():
            d = os.path.join(directory, name)
            os.makedirs(d, exist_ok=True)
            ckpt = {"state_dict": OrderedDict((k, v.detach().cpu()) for k, v in self._models[name].state_dict().items()),
                    "epoch": epoch + 1, "optimizer": {"momentum": mom.detach().cpu(), "scalars": scal.cpu()},
                    "scheduler": {"last_epoch": self.sched.last_epoch}}
            path = os.path.join(d, model_name or f"model.pth.tar-{epoch + 1}")
            torch.save(ckpt, path)
            if is_best:
                torch.save(ckpt, os.path.join(d, "model-best.pth.tar"))

    def load_model(self, directory, epoch=None):
        """trainers/GLP_OT_SVLoRA.py:1023-1053: `<directory>/<name>/model-best.pth.tar` (or `model.pth.tar-<epoch>`) for
        every registered model; the fixed token vectors are ignored; load_state_dict(strict=False)."""
        if not directory:
            print("Note that load_model() is skipped as no pretrained model is given")
            return
        model_file = "model-best.pth.tar" if epoch is None else "model.pth.tar-" + str(epoch)
        frozen_changed = False
        train = set(self.engine.params.keys)
        for name in self.get_model_names():
            path = os.path.join(directory, name, model_file)
            if not os.path.exists(path):
                raise FileNotFoundError('Model not found at "{}"'.format(path))
            ckpt = torch.load(path, map_location="cpu")
            sd = ckpt["state_dict"]
            for k in ("token_prefix", "token_suffix"):
                sd.pop(k, None)
            print('Loading weights to {} from "{}" (epoch = {})'.format(name, path, ckpt["epoch"]))
            self._models[name].load_state_dict(sd, strict=False)
            frozen_changed |= any(f"{name}.{k}" not in train for k in sd)
        if frozen_changed:                                            # rebuild the compute-dtype copies of frozen tensors
            self.engine.load_frozen(self.model.state_dict())

    def fed_before_train(self, is_global=False):
        self.start_epoch = 0
        self.total_time_start = time.time()

    def fed_after_train(self):
        print(f"Total time Elapsed: {round(time.time() - self.total_time_start)}s")

    # -------------------------------------------------------------- test --
    @torch.no_grad()
    def test(self, split=None, is_global=False, current_epoch=0, idx=-1, global_test=False):
        """[acc, err, macro_f1, auc] over the client's test loader, all four in percent as the reference's evaluator
        stores them (SimpleTrainer.test, Dassl/dassl/engine/trainer.py:523-569; evaluation/evaluator_oph.py:66-96:
        auc = 100 * compute_auc; federated_main.py:685-690 indexes [0..3])."""
        self.set_model_mode("eval")
        probs, labels, attrs_all = [], [], []
        for batch in self.fed_test_loader_x_dict[idx]:
            image, label, attrs, attr = self.parse_batch_test(batch)
            logits = self.model_inference(image, attr)
            probs.append(torch.softmax(logits, -1))
            labels.append(label)
            attrs_all.append(attrs)
        prob_d, y_d = torch.cat(probs).float().contiguous(), torch.cat(labels).contiguous()
        attrs_d = torch.cat(attrs_all, dim=1)                                  # [n_attr, N]
        if prob_d.shape[1] == 2 and not getattr(self.cfg.TEST, "HOST_METRICS", False):
            # binary task: one pass over the scores on the GPU yields the integer counts every reported score is a
            # ratio of (csrc/evalmetrics.hip); one small D2H copy per test() instead of the whole score matrix
            G = 8                                                              # FFM_MAX_GROUPS: absent groups are skipped
            tables = torch.stack([ops.eval_counts(prob_d, y_d, attrs_d[a].contiguous(), G)
                                  for a in range(attrs_d.shape[0])]).cpu().numpy()
            res = basic_from_counts(tables[0])
            self.last_results = {"accuracy": res[0], "error_rate": res[1]}
            if tables[0][-1][0] > 0 and tables[0][-1][1] > 0:
                self.last_results.update(comprehensive_scores_from_counts(tables))
            return res
        prob = prob_d.cpu().numpy()
        y = y_d.cpu().numpy()
        pred = prob.argmax(-1)
        acc = 100.0 * float((pred == y).mean())
        # the fairness block of Classification_oph.evaluate (evaluation/evaluator_oph.py:69-113), binary tasks
        self.last_results = {"accuracy": acc, "error_rate": 100.0 - acc}
        if prob.shape[1] == 2 and y.min() != y.max():
            self.last_results.update(comprehensive_scores(prob, y, attrs_d.cpu().numpy()))
        return [acc, 100.0 - acc, 100.0 * macro_f1(pred, y, prob.shape[1]), 100.0 * auc_macro_ovr(prob, y)]
