"""Federated round loop of the FairLoRA runs: the ``FedOTPLoRA`` branch of the reference's
``federated_main.py`` (:604-726) and ``average_weights_EMA`` (utils/fed_utils.py:42-100) for a list of clients.

SURVEY §8(f) rank 1.  Same flag names as the reference's argparse (``--num_users --frac --round --avg_prompt
--num_prompt --idxs_users_train --idxs_users_test --shared_half_s``), carried in ``FedArgs``.  The loop is control
plane: it sequences ``trainer.train / trainer.test / model.state_dict / load_state_dict(strict=False)`` and
averages a few hundred KB of trainable tensors per round; the local training it calls is the HIP engine.

Two drivers:
  * ``run_fedotplora``        one process, clients trained one after the other on one GPU (the reference's shape);
  * ``run_fedotplora_ranks``  one process per GPU under torch.distributed: the round's clients are dealt
    round-robin to the ranks, every rank trains its share, and the weighted sum is ONE all-reduce of the flat
    trainable buffer (fedavg.element_weights gives the reference's per-element weights).
"""
from __future__ import annotations

import copy
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .fedavg import is_group_s_block

Tensor = torch.Tensor


@dataclass
class FedArgs:
    """The reference's CLI flags that the FedOTPLoRA branch reads (federated_main.py:791-881)."""
    num_users: int
    frac: float = 1.0
    round: int = 50
    avg_prompt: int = 1
    num_prompt: int = 2
    idxs_users_train: List[int] = field(default_factory=list)
    idxs_users_test: List[int] = field(default_factory=list)
    shared_half_s: bool = False
    local_s: bool = False                 # cfg.TRAINER.GLP_OT_LORA.LOCAL_S
    seed: Optional[int] = None            # seeds numpy's global generator like the reference's set_random_seed
    # run_fedotplora_ranks only.  The reference trains its clients one after the other with ONE optimizer and ONE
    # LR scheduler (federated_main.py:183): momentum buffers and the StepLR counter leak from
    # client to client.  True reproduces that across ranks (clients of a round train in reference order, the
    # optimizer state is handed on after each: no concurrency, for parity runs); False gives every rank its own
    # optimizer state (clients of a round train concurrently: the throughput mode).
    compat_sequential_optimizer: bool = False


def average_weights_ema(w_g: Dict[str, Tensor], w: Dict[int, Dict[str, Tensor]], idxs_users: Sequence[int],
                        datanumber_client: Sequence[int], datanumber_client_by_attr, epoch: int, max_epoch: int,
                        beta: float = 0.999, shared_half_s: bool = False) -> Dict[str, Tensor]:
    """Weighted FedAvg + EMA with the previous global weights (utils/fed_utils.py:42-100).  Entries are weighted
    by n_k / sum n; row g of every ``lora_S`` [G, r] tensor by n_{k,g} / sum_k n_{k,g}; optional column mean over the
    groups on the first half of the rank ("shared_half_s"); then (1 - b) avg + b w_g with b = beta * epoch / max_epoch."""
    idxs_users = [int(u) for u in idxs_users]
    total = sum(datanumber_client[u] for u in idxs_users)
    by_attr = tot_by_attr = None
    if datanumber_client_by_attr is not None:
        by_attr = torch.tensor(datanumber_client_by_attr)
        tot_by_attr = by_attr[idxs_users].sum(0)
    G = None if by_attr is None else by_attr.shape[1]
    beta_decay = beta * (epoch / max(max_epoch, 1))
    out: Dict[str, Tensor] = {}
    for key, first in w[idxs_users[0]].items():
        # (2-D blocks only: a 1-D lora_S_global / SVLoRA lora_S [r] with r == G would be broadcast to [G, r] by the
        # reference's shape[0] test - a latent shape bug there, not a behaviour to reproduce)
        grouped = by_attr is not None and is_group_s_block(key, first.shape, G)
        acc = None
        for u in idxs_users:
            x = w[u][key]
            if grouped:
                term = x * (by_attr[u] / tot_by_attr)[:, None].to(x.device)
            else:
                term = x * (datanumber_client[u] / total)
            acc = term if acc is None else acc + term
        if shared_half_s and grouped:
            n_groups, n_dim = acc.shape
            acc = torch.cat([acc[:, : n_dim // 2].mean(0, keepdim=True).repeat(n_groups, 1), acc[:, n_dim // 2:]], dim=1)
        out[key] = (1 - beta_decay) * acc + beta_decay * w_g[key]
    return out


def average_weights(w, idxs_users: Sequence[int], datanumber_client: Sequence[int], datanumber_client_by_attr=None,
                    islist: bool = False):
    """Plain weighted FedAvg without the EMA (utils/fed_utils.py:6-40): entries weighted by n_k / sum n, row g of every
    per-group ``lora_S`` [G, r] block by n_{k,g} / sum_k n_{k,g}.  ``islist``: ``w[u]`` is one tensor per client (the
    reference's prompt-only trainers pass ``ctx`` this way) instead of a state_dict.  Equal to ``average_weights_ema``
    at ``epoch = 0`` without ``shared_half_s`` - kept under the reference's name and signature."""
    idxs_users = [int(u) for u in idxs_users]
    total = sum(datanumber_client[u] for u in idxs_users)
    if islist:
        acc = None
        for u in idxs_users:
            term = w[u] * (datanumber_client[u] / total)
            acc = term if acc is None else acc + term
        return acc
    w_g = {k: torch.zeros_like(v) for k, v in w[idxs_users[0]].items()}
    return average_weights_ema(w_g, w, idxs_users, datanumber_client, datanumber_client_by_attr, 0, 1,
                               shared_half_s=False)


def select_clients(epoch: int, args: FedArgs, dataset_users: int) -> List[int]:
    """federated_main.py:606-613: an explicit list wins; round 0 trains every client; later rounds draw
    max(int(frac * num_users), 1) of them without replacement from numpy's global generator."""
    if len(args.idxs_users_train) > 0:
        return list(args.idxs_users_train)
    if epoch == 0:
        return list(range(dataset_users))
    m = max(int(args.frac * args.num_users), 1)
    return [int(i) for i in np.random.choice(range(args.num_users), m, replace=False)]


def personalize(global_weights: Dict[str, Tensor], idx: int, args: FedArgs, local_ctx: Dict[int, Tensor],
                local_s: Dict[int, Dict[str, Tensor]]) -> Dict[str, Tensor]:
    """federated_main.py:645-652: a client listed in --idxs_users_train keeps its own local prompts
    ctx[avg_prompt:num_prompt] and, under LOCAL_S, its own lora_S tensors; everything else is the new global."""
    w = copy.deepcopy(global_weights)
    if idx in args.idxs_users_train:
        w["prompt_learner.ctx"][args.avg_prompt:args.num_prompt] = local_ctx[idx]
        if args.local_s:
            for k, v in local_s[idx].items():
                w[k] = v
    return w


def _counts(trainer, attribute: str, n_users: int):
    n_client = [len(trainer.fed_train_loader_x_dict[i].dataset) for i in range(n_users)]
    by_attr = [trainer.fed_train_loader_x_dict[i].dataset.count_by_attribute(attribute) for i in range(n_users)]
    return n_client, by_attr


def run_fedotplora(trainer, args: FedArgs, attribute: Optional[str] = None, log=print) -> Dict[str, list]:
    """One process, one GPU: the reference's round loop verbatim in structure.  Returns the per-round means of
    [accuracy, error_rate, macro_f1, auc] over the tested clients, and the final per-client weights."""
    cfg = trainer.cfg
    users = cfg.DATASET.USERS if hasattr(cfg.DATASET, "USERS") else args.num_users
    attribute = attribute or cfg.DATASET.ATTRIBUTE_TYPE
    if args.seed is not None:
        np.random.seed(args.seed)
    n_client, by_attr = _counts(trainer, attribute, users)
    trainer.fed_before_train()
    # Only the trainable tensors travel: the reference averages the whole state_dict, but frozen tensors are
    # identical on every client, so their weighted mean is the tensor itself up to one rounding per round
    # (SURVEY §5 quirk 10); leaving them out also keeps the engine's packed copies of the frozen weights valid.
    # RN50: the BatchNorm running statistics do differ between clients and are averaged with everything else.
    keys = set(trainer.engine.params.keys) if hasattr(trainer, "engine") else None
    if keys is not None and hasattr(trainer.engine, "buffer_views"):
        keys |= set(trainer.engine.buffer_views())
    snap = lambda: copy.deepcopy({k: v for k, v in trainer.model.state_dict().items() if keys is None or k in keys})
    global_weights = snap()
    local_weights: Dict[int, Dict[str, Tensor]] = {}
    local_weights_per = {i: copy.deepcopy(global_weights) for i in range(users)}
    local_ctx: Dict[int, Tensor] = {}
    local_s: Dict[int, Dict[str, Tensor]] = {}
    hist = {"acc": [], "err": [], "f1": [], "auc": [], "epoch": [], "time": []}
    start = time.time()
    for epoch in range(args.round):
        idxs_users = select_clients(epoch, args, users)
        log(f"------------local train start epoch: {epoch} -------------")
        for idx in idxs_users:
            trainer.model.load_state_dict(global_weights if epoch == 0 else local_weights_per[idx], strict=False)
            trainer.train(idx=idx, global_epoch=epoch, is_fed=True, is_last_client=idx == idxs_users[-1])
            lw = snap()
            local_ctx[idx] = copy.deepcopy(lw["prompt_learner.ctx"][args.avg_prompt:args.num_prompt])
            local_s[idx] = copy.deepcopy({k: v for k, v in lw.items() if "lora_S" in k})
            local_weights[idx] = lw
        global_weights = average_weights_ema(global_weights, local_weights, idxs_users, n_client, by_attr, epoch,
                                             args.round, shared_half_s=args.shared_half_s)
        all_users = list(args.idxs_users_test) if len(args.idxs_users_test) > 0 else list(range(users))
        results = []
        for idx in all_users:
            local_weights_per[idx] = personalize(global_weights, idx, args, local_ctx, local_s)
            trainer.model.load_state_dict(local_weights_per[idx], strict=False)
            results.append(trainer.test(idx=idx, current_epoch=epoch))
        for name, col in (("acc", 0), ("err", 1), ("f1", 2)):
            hist[name].append(sum(r[col] for r in results) / len(results))
        if len(results[0]) > 3:
            hist["auc"].append(sum(r[3] for r in results) / len(results))
        hist["epoch"].append(epoch)
        hist["time"].append(time.time() - start)
        log(f"Global test acc: {hist['acc'][-1]}  macro_f1: {hist['f1'][-1]}"
            + (f"  auc: {hist['auc'][-1]}" if hist["auc"] else ""))
    trainer.fed_after_train()
    hist["global_weights"] = global_weights
    hist["local_weights_per"] = local_weights_per
    return hist


def run_fedotplora_ranks(trainer, args: FedArgs, attribute: Optional[str] = None, log=print) -> Dict[str, list]:
    """One process per GPU.  Round r: client idxs_users[j] trains on rank j % world; every rank then contributes
    sum_k w_k (.) theta_k of ITS clients to one all-reduce of the flat trainable buffer (RCCL over xGMI, gloo on
    CPU), and finishes shared_half_s + EMA locally.  Frozen tensors are identical on all ranks and stay put.

    The round boundary IS ``fedavg.FedAvgAggregator`` (utils/fed_utils.py:42-100): device-resident cached weight vectors,
    ffm_scale_by / ffm_scale_acc -> one all_reduce -> ffm_fedavg_finish on the GPU - the same object ``bench.py --gpus N``
    times.  RN50's BatchNorm buffers go through a second instance of it (plain n_k / sum n weights, no lora_S blocks)."""
    from .fedavg import FedAvgAggregator
    assert dist.is_initialized(), "launch under torch.distributed.run"
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = trainer.cfg
    users = cfg.DATASET.USERS if hasattr(cfg.DATASET, "USERS") else args.num_users
    attribute = attribute or cfg.DATASET.ATTRIBUTE_TYPE
    if args.seed is not None:
        np.random.seed(args.seed)
    n_client, by_attr = _counts(trainer, attribute, users)
    params = trainer.engine.params
    flat, offsets = params.flat, params.offsets
    lo = trainer.engine.cfg.lora
    trainer.fed_before_train()
    agg = FedAvgAggregator(flat, offsets, lo.num_groups, lo.rank, shared_half_s=args.shared_half_s, beta=0.999)
    global_flat = agg.global_prev                                    # updated in place by agg.finish()
    per_client = {i: global_flat.clone() for i in range(users)}      # personalised flat buffers (this rank's view)
    ctx_off, ctx_shape = offsets["prompt_learner.ctx"]
    n_ctx_row = int(np.prod(ctx_shape[1:]))
    lo_a, lo_b = ctx_off + args.avg_prompt * n_ctx_row, ctx_off + args.num_prompt * n_ctx_row
    s_slices = [(off, off + int(np.prod(shp))) for k, (off, shp) in offsets.items() if "lora_S" in k]
    hist = {"acc": [], "err": [], "f1": [], "auc": [], "epoch": []}
    # RN50: the BatchNorm running statistics (buffers, not parameters) are averaged like every other state_dict
    # entry, with the plain n_k / sum n weights; they ride in a second, small all-reduce
    eng = trainer.engine
    has_buf = hasattr(eng, "buffers_flat")
    agg_buf = FedAvgAggregator(eng.buffers_flat(), {}, lo.num_groups, lo.rank, shared_half_s=False, beta=0.999) if has_buf else None
    global_buf = agg_buf.global_prev if has_buf else None
    per_client_buf = {i: global_buf.clone() for i in range(users)} if has_buf else None
    # LR schedule: the reference's ONE scheduler advances by (local epochs x registered names) per trained client, in
    # client order, whichever process trains it.  Every rank therefore positions its scheduler from the GLOBAL count of
    # client-epochs before each of its clients (and after the round), so the schedule does not depend on the world size
    can_position = hasattr(trainer, "set_lr_epoch") and not args.compat_sequential_optimizer
    per_client_epochs = trainer.max_epoch * trainer.steps_per_update() if can_position else 0
    sched_pos = trainer.sched.last_epoch if can_position else 0
    for epoch in range(args.round):
        # rank 0 draws the round's clients (numpy's global generator, like the reference) and everybody uses ITS list:
        # with an unseeded generator (--seed <= 0) the ranks would otherwise pick different subsets and the all-reduce
        # would silently combine inconsistent participants
        pick = [select_clients(epoch, args, users) if rank == 0 else None]
        dist.broadcast_object_list(pick, src=0)
        idxs_users = [int(u) for u in pick[0]]
        agg.begin()
        if has_buf:
            agg_buf.begin()
        local_after: Dict[int, Tensor] = {}
        for j, idx in enumerate(idxs_users):
            mine = j % world == rank
            if mine:
                flat.copy_(global_flat if epoch == 0 else per_client[idx])
                if has_buf:
                    eng.load_buffers_flat(global_buf if epoch == 0 else per_client_buf[idx])
                if can_position:
                    trainer.set_lr_epoch(sched_pos + j * per_client_epochs)
                trainer.train(idx=idx, global_epoch=epoch, is_fed=True, is_last_client=idx == idxs_users[-1])
                if args.idxs_users_train:
                    local_after[idx] = flat.detach().clone()
                agg.add(flat, idx, idxs_users, n_client, by_attr)
                if has_buf:
                    agg_buf.add(eng.buffers_flat(), idx, idxs_users, n_client, None)
            if args.compat_sequential_optimizer and world > 1:
                # hand the shared optimizer on: the next client (on whichever rank) starts from this one's momentum
                # buffers, first-step flag, StepLR counter and learning rate
                mom, scal = trainer.optimizer_state()
                dist.broadcast(mom, src=j % world)
                dist.broadcast(scal, src=j % world)
                if not mine:
                    trainer.load_optimizer_state(mom, scal)
        if can_position:
            sched_pos += len(idxs_users) * per_client_epochs
            trainer.set_lr_epoch(sched_pos)
        # one all_reduce(SUM) of the flat buffer, shared_half_s under the reference's guard (utils/fed_utils.py:90),
        # EMA with the previous global: global_flat / global_buf are updated in place
        agg.finish(epoch, args.round, grouped=by_attr is not None)
        if has_buf:
            agg_buf.finish(epoch, args.round, grouped=False)
        # personalisation needs every trained client's local prompts / lora_S on every rank that may test it:
        # exchange them (a few KB) with one more all-reduce of a zero-padded buffer
        keep = torch.zeros(users, flat.numel(), device=flat.device) if args.idxs_users_train else None
        if keep is not None:
            for idx, f in local_after.items():
                keep[idx] = f
            dist.all_reduce(keep, op=dist.ReduceOp.SUM)
        all_users = list(args.idxs_users_test) if len(args.idxs_users_test) > 0 else list(range(users))
        results = []
        for idx in all_users:
            per_client[idx] = global_flat.clone()
            if has_buf:
                per_client_buf[idx] = global_buf.clone()
            if idx in args.idxs_users_train and keep is not None:
                per_client[idx][lo_a:lo_b] = keep[idx][lo_a:lo_b]
                if args.local_s:
                    for a, b in s_slices:
                        per_client[idx][a:b] = keep[idx][a:b]
        for j, idx in enumerate(all_users):                          # evaluation is dealt to the ranks as well
            if j % world != rank:
                continue
            flat.copy_(per_client[idx])
            if has_buf:
                eng.load_buffers_flat(per_client_buf[idx])
            results.append((idx, trainer.test(idx=idx, current_epoch=epoch)))
        gathered = [None] * world
        dist.all_gather_object(gathered, results)
        flat_res = [r for part in gathered for r in part]
        for name, col in (("acc", 0), ("err", 1), ("f1", 2), ("auc", 3)):
            vals = [r[1][col] for r in flat_res if len(r[1]) > col]
            if vals:
                hist[name].append(sum(vals) / len(vals))
        hist["epoch"].append(epoch)
        if rank == 0:
            log(f"round {epoch}: acc {hist['acc'][-1]:.3f} f1 {hist['f1'][-1]:.3f}"
                + (f" auc {hist['auc'][-1]:.4f}" if hist["auc"] else ""))
    flat.copy_(global_flat)
    if has_buf:
        eng.load_buffers_flat(global_buf)
        hist["global_buffers"] = global_buf.clone()
    trainer.fed_after_train()
    hist["global_flat"] = global_flat.clone()
    # the per-client weights the reference saves as global_client{idx}_final.pth (federated_main.py:771-774), rebuilt
    # from the flat buffers (every rank holds all of them: personalised pieces were exchanged above)
    lwp = {}
    for idx, f in per_client.items():
        w = {k: f[off:off + int(np.prod(shp))].view(shp).clone() for k, (off, shp) in offsets.items()}
        if has_buf:                                                  # layout of RN50Engine.buffers_flat()
            off = 0
            buf = per_client_buf[idx]
            for name in ("running_mean", "running_var"):
                for bn in eng.bns:
                    w[bn.prefix + name] = buf[off:off + bn.C].clone()
                    off += bn.C
            for i, bn in enumerate(eng.bns):
                w[bn.prefix + "num_batches_tracked"] = buf[off + i].to(torch.int64)
        lwp[idx] = w
    hist["local_weights_per"] = lwp
    return hist
