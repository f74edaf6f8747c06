"""Command line of the FairLoRA federated run, flag names as ``federated_main.py:791-881``:

    python -m fairfedmed_amd.federated_main --model FedOTPLoRA --trainer GLP_OT_SVLoRA --root DATA/ \
        --num_users 3 --frac 0.8 --round 50 --stepsize 200 --lr 0.001 --gamma 0.1 --OT None \
        --attribute_type language --n_ctx 4 --num_prompt 2 --unfreeze_image_encoder True \
        --lora_rank 12 --lora_alpha 2 --lora_type FairLoRA --shared_half_s True \
        --dataset-config-file configs/datasets/fairfedmed.yaml --config-file configs/trainers/GLP_OT/vit_b16_oph.yaml \
        --output-dir output/run1

One process: the reference's sequential round loop (``federated.run_fedotplora``).  Under
``python -m torch.distributed.run --nproc-per-node N`` the clients of a round are dealt to the ranks and the round
ends in one all-reduce (``federated.run_fedotplora_ranks``; ``--compat-sequential-optimizer`` restores the
reference's shared optimizer).  Only the branch the FairLoRA scripts use is built: ``--model FedOTPLoRA``,
``--trainer GLP_OT_SVLoRA`` (``--OT None | Sinkhorn | COT``); anything else raises NotImplementedError.

Differences a user must know (no network in the build image): pretrained CLIP weights are not downloaded - pass
``--state-dict file.pt`` (a CustomCLIP state_dict) or the deterministic synthetic weights are used; ``--synthetic``
replaces the dataset by synthetic batches of the same shapes.
"""
from __future__ import annotations

import argparse
import ast
import os
import sys
from types import SimpleNamespace as NS
from typing import List, Optional

import numpy as np
import torch

from . import config as C

_3D = ("oct_bscans", "oct_bscans_3d", "mac_onh", "onh_mac")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="fairfedmed_amd.federated_main")
    a = p.add_argument
    a("--model", type=str, default="FedOTPLoRA")
    a("--trainer", type=str, default="GLP_OT_SVLoRA")
    a("--round", type=int, default=10)
    a("--stepsize", type=int, default=-1)
    a("--num_users", type=int, default=10)
    a("--frac", type=float, default=1)
    a("--lr", type=float, default=0.001)
    a("--gamma", type=float, default=0.1)
    a("--train_batch_size", type=int, default=32)
    a("--test_batch_size", type=int, default=100)
    a("--seed", type=int, default=1)
    a("--attribute_type", type=str, default="race")
    a("--attributes", type=str, nargs="+", default=["gender", "race", "ethnicity", "language", "maritalstatus"])
    a("--modality_type", type=str, default="slo_fundus")
    a("--dim_per_3d_slice", type=int, default=16)
    # type=bool on purpose: like the reference, ANY non-empty value parses to True, "False" included
    a("--input_no_transform", type=bool, default=False)
    a("--n_ctx", type=int, default=16)
    a("--num_prompt", type=int, default=2)
    a("--avg_prompt", type=int, default=1)
    a("--ctx_init", default=False)
    a("--OT", type=str, default="COT")
    a("--top_percent", type=float, default=1)
    a("--eps", type=float, default=0.1)
    a("--thresh", type=float, default=1e-3)
    a("--max_iter", type=int, default=100)
    a("--unfreeze_image_encoder", type=bool, default=False)
    a("--unfreeze_text_encoder", type=bool, default=False)
    a("--lora_rank", type=int, default=4)
    a("--lora_alpha", type=float, default=0.04)
    a("--lora_type", type=str, default="LoRA")
    a("--lora_local_s", type=bool, default=False)
    a("--shared_half_s", type=bool, default=False)
    a("--lora_global_s", type=bool, default=False)
    a("--lambda_fairness", type=float, default=0.0)
    a("--idxs_users_train", type=list, default=[])
    a("--idxs_users_test", type=list, default=[])
    a("--disable_attr", action="store_true")
    a("--root", type=str, default="/DATA/")
    a("--output-dir", type=str, default="output/..")
    a("--config-file", type=str, default="")
    a("--dataset-config-file", type=str, default="")
    a("--backbone", type=str, default="")
    a("--eval-only", action="store_true")
    a("--model-dir", type=str, default="")
    a("--load-epoch", type=int)
    # accepted for script compatibility, unused on this path
    for flag, kw in (("--partition", dict(type=str, default="noniid-labeldir100")), ("--beta", dict(type=float, default=0.1)),
                     ("--iid", dict(default=False)), ("--useall", dict(default=False)), ("--num_shots", dict(type=int, default=2)),
                     ("--mu", dict(type=float, default=0.5)), ("--logdir", dict(type=str, default="./logs/"))):
        a(flag, **kw)
    # build-specific
    a("--prec", type=str, default="bf16", choices=["bf16", "fp32", "amp", "fp16"],
      help="TRAINER.GLP_OT.PREC; the reference's 'fp16' runs as bf16 here, with a warning (trainer.resolve_precision)")
    a("--state-dict", type=str, default="", help="CustomCLIP state_dict (.pt) with the pretrained CLIP weights")
    a("--synthetic", action="store_true", help="synthetic batches instead of the dataset under --root")
    a("--synthetic-batches", type=int, default=4)
    a("--transport", type=str, default="uint8", choices=["uint8", "float32"])
    a("--compat-sequential-optimizer", action="store_true")
    a("--save-trainable-only", action="store_true",
      help="global_client{idx}_final.pth without the frozen CLIP tensors (the reference saves the full state_dict)")
    return p


def _merge_yaml(cfg: NS, path: str) -> None:
    """yacs' merge_from_file for the handful of keys the two config files set (nested mappings, tuples as text)."""
    import yaml
    with open(path) as f:
        tree = yaml.safe_load(f) or {}

    def merge(node: NS, d: dict):
        for k, v in d.items():
            if isinstance(v, dict):
                if not isinstance(getattr(node, k, None), NS):
                    setattr(node, k, NS())
                merge(getattr(node, k), v)
            else:
                if isinstance(v, str) and v[:1] in "([":
                    v = ast.literal_eval(v)
                setattr(node, k, v)
    merge(cfg, tree)


def setup_cfg(args) -> NS:
    """extend_cfg + setup_cfg (federated_main.py:30-160) as a SimpleNamespace tree: defaults, then the dataset config
    file, then the method config file, then the command line."""
    cfg = NS(
        SEED=args.seed, OUTPUT_DIR=args.output_dir, VERBOSE=True,
        INPUT=NS(SIZE=(224, 224), PIXEL_MEAN=list(C.CLIP_PIXEL_MEAN), PIXEL_STD=list(C.CLIP_PIXEL_STD),
                 NO_TRANSFORM=args.input_no_transform),
        DATASET=NS(NAME="FairFedMed", ROOT=args.root, USERS=args.num_users, ATTRIBUTE_TYPE=args.attribute_type,
                   ATTRIBUTES=list(args.attributes), MODALITY_TYPE=args.modality_type,
                   DIM_PER_3D_SLICE=args.dim_per_3d_slice),
        DATALOADER=NS(TRAIN_X=NS(BATCH_SIZE=args.train_batch_size), TEST=NS(BATCH_SIZE=args.test_batch_size), NUM_WORKERS=0),
        MODEL=NS(BACKBONE=NS(NAME="ViT-B/16", PRETRAINED=True), STATE_DICT=None),
        OPTIM=NS(NAME="sgd", LR=args.lr, MOMENTUM=0.9, WEIGHT_DECAY=5e-4, LR_SCHEDULER="single_step",
                 STEPSIZE=args.stepsize, GAMMA=args.gamma, MAX_EPOCH=1, ROUND=args.round),
        TRAIN=NS(CHECKPOINT_FREQ=0, PRINT_FREQ=10, METRICS_EVERY=1),
        TEST=NS(BATCH_SIZE=args.test_batch_size, NO_TEST=False, EVALUATOR="Classification_oph"),
        TRAINER=NS(NAME=args.trainer, LAMBDA_FAIRNESS=args.lambda_fairness,
                   GLP_OT=NS(N_CTX=args.n_ctx, CSC=False, CTX_INIT=args.ctx_init, PREC=args.prec,
                             CLASS_TOKEN_POSITION="end", N=args.num_prompt, THRESH=args.thresh, EPS=args.eps, OT=args.OT,
                             TOP_PERCENT=args.top_percent, MAX_ITER=args.max_iter),
                   GLP_OT_LORA=NS(UNFREEZE_IMAGE_ENCODER=args.unfreeze_image_encoder,
                                  UNFREEZE_TEXT_ENCODER=args.unfreeze_text_encoder, RANK=args.lora_rank,
                                  ALPHA=args.lora_alpha, TYPE=args.lora_type, LOCAL_S=args.lora_local_s,
                                  GLOBAL_S=args.lora_global_s, DISABLE_ATTR=args.disable_attr)),
    )
    for path in (args.dataset_config_file, args.config_file):
        if path:
            _merge_yaml(cfg, path)
    # the command line wins over the files for these (federated_main.py:30-58)
    cfg.OPTIM.LR, cfg.OPTIM.STEPSIZE, cfg.OPTIM.ROUND, cfg.OPTIM.GAMMA, cfg.OPTIM.MAX_EPOCH = \
        args.lr, args.stepsize, args.round, args.gamma, 1
    cfg.DATALOADER.TRAIN_X.BATCH_SIZE = getattr(cfg.DATALOADER.TRAIN_X, "BATCH_SIZE", args.train_batch_size)
    cfg.TEST.BATCH_SIZE = getattr(getattr(cfg.DATALOADER, "TEST", NS()), "BATCH_SIZE", args.test_batch_size)
    if args.backbone:
        cfg.MODEL.BACKBONE.NAME = args.backbone
    if args.trainer:
        cfg.TRAINER.NAME = args.trainer
    return cfg


def check_scope(args, cfg) -> None:
    if args.model != "FedOTPLoRA":
        raise NotImplementedError(f"--model {args.model}: only the FedOTPLoRA branch (federated_main.py:604-726) is built")
    if cfg.TRAINER.NAME != "GLP_OT_SVLoRA":
        raise NotImplementedError(f"--trainer {cfg.TRAINER.NAME}: only GLP_OT_SVLoRA is built")
    if str(args.OT) not in ("None", "Sinkhorn", "COT"):
        raise NotImplementedError(f"--OT {args.OT}: choose None (the FairLoRA scripts), Sinkhorn or COT")
    if not args.unfreeze_image_encoder:
        raise NotImplementedError("--unfreeze_image_encoder must be set: without it no FairLoRA adapter is injected")


def main(argv: Optional[List[str]] = None, log=print, cfg_hook=None):
    """cfg_hook(cfg): last-minute edits of the config tree (tests use it to select a reduced model geometry)."""
    import torch.distributed as dist
    from . import federated as F
    from .registry import build_trainer
    from . import trainer as _trainer  # noqa: F401  (registers GLP_OT_SVLoRA)
    args = build_parser().parse_args(argv)
    args.idxs_users_train = [int(i) for i in args.idxs_users_train]
    args.idxs_users_test = [int(i) for i in args.idxs_users_test]
    for i in args.idxs_users_train + args.idxs_users_test:
        assert i < args.num_users, "idx of users must be less than num_users"
    cfg = setup_cfg(args)
    check_scope(args, cfg)
    ranks = int(os.environ.get("WORLD_SIZE", "1")) > 1 or "RANK" in os.environ
    if ranks and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("FFM_ONE_DEVICE"):                # test rig: every rank on cuda:0, gloo instead of RCCL
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cfg.DEVICE = f"cuda:{torch.cuda.current_device()}"       # one client per GPU: this rank's device
    if args.seed > 0:                                       # set_random_seed (Dassl/dassl/utils/tools.py)
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    is3d = cfg.DATASET.MODALITY_TYPE in _3D
    if args.synthetic:
        base = C.rn50() if cfg.MODEL.BACKBONE.NAME in ("RN50", "rn50") else C.vit_b16()
        from .trainer import ATTRIBUTE_GROUPS
        G = len(ATTRIBUTE_GROUPS[cfg.DATASET.NAME][args.attribute_type])
        mcfg = C.ModelCfg(vision=base.vision, text=base.text, lora=C.LoraCfg(rank=args.lora_rank, alpha=args.lora_alpha,
                                                                            num_groups=G),
                          n_prompts=args.num_prompt, n_ctx=args.n_ctx,
                          dim_per_3d_slice=args.dim_per_3d_slice if is3d else 0)
        cfg.DATASET.ATTRIBUTES = [args.attribute_type]
        cfg.DATA = _trainer.SyntheticFedData(mcfg, args.num_users, args.synthetic_batches, 1,
                                             cfg.DATALOADER.TRAIN_X.BATCH_SIZE, attribute=args.attribute_type,
                                             signal=0.3)
    else:
        from .data import FedData
        cfg.DATA = FedData(cfg, transport=args.transport)
    if args.state_dict:
        cfg.MODEL.STATE_DICT = torch.load(args.state_dict, map_location="cpu")
    else:
        log("NOTE: no --state-dict given and CLIP weights cannot be downloaded here: deterministic synthetic weights")
    if cfg.MODEL.BACKBONE.NAME in ("tiny", "rn_tiny"):      # reduced geometries for smoke runs (64 x 64 inputs)
        cfg.MODEL.GEOMETRY = C.vit_tiny() if cfg.MODEL.BACKBONE.NAME == "tiny" else C.rn_tiny()
    if cfg_hook is not None:
        cfg_hook(cfg)
    tr = build_trainer(cfg)
    if args.eval_only:                                      # federated_main.py: load_model(model_dir, load_epoch), test, stop
        tr.load_model(args.model_dir, epoch=args.load_epoch)
        users = list(args.idxs_users_test) or list(range(args.num_users))
        results = {idx: tr.test(idx=idx) for idx in users}
        for idx, r in results.items():
            log(f"client {idx}: acc {r[0]:.3f} err {r[1]:.3f} macro_f1 {r[2]:.3f} auc {r[3]:.4f}")
        return {"eval": results}
    fargs = F.FedArgs(num_users=args.num_users, frac=args.frac, round=args.round, avg_prompt=args.avg_prompt,
                      num_prompt=args.num_prompt, idxs_users_train=args.idxs_users_train,
                      idxs_users_test=args.idxs_users_test, shared_half_s=args.shared_half_s, local_s=args.lora_local_s,
                      seed=args.seed if args.seed > 0 else None,
                      compat_sequential_optimizer=args.compat_sequential_optimizer)
    rank = dist.get_rank() if ranks else 0
    hist = (F.run_fedotplora_ranks if ranks else F.run_fedotplora)(tr, fargs, log=log if rank == 0 else (lambda *_: None))
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        # federated_main.py:771-774: one global_client{idx}_final.pth per client holding the client's FULL state_dict
        # (the frozen CLIP tensors are the same in every file; --save-trainable-only keeps the tensors that differ)
        base = None if args.save_trainable_only else {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
        for idx, w in hist["local_weights_per"].items():
            name = os.path.join(args.output_dir, f"global_client{idx}_final.pth")
            log(f"Save client-{idx} global weights: {name}")
            sd = {k: v.detach().cpu() for k, v in w.items()}
            if base is not None:
                sd = type(base)((k, sd.get(k, v)) for k, v in base.items())
            torch.save(sd, name)
        log("global_test_acc_list:", hist["acc"])
        log("maximum test acc:", max(hist["acc"]))
        log("mean of acc:", float(np.mean(hist["acc"][-5:])))
        log("std of acc:", float(np.std(hist["acc"][-5:])))
        if hist.get("auc"):
            log("global_test_auc_list:", hist["auc"])
    if ranks:
        dist.destroy_process_group()
    return hist


if __name__ == "__main__":
    main(sys.argv[1:])
