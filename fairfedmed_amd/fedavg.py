"""Round boundary: weighted FedAvg + EMA of the trainable tensors as ONE
all-reduce of the flat parameter buffer.

Restates ``average_weights_EMA`` (utils/fed_utils.py:42-100) for one client per
rank: every rank pre-multiplies its flat buffer by its own weights
(n_k / sum n for ordinary entries, n_{k,g} / sum_k n_{k,g} for row g of every
lora_S block), one all_reduce(SUM) over RCCL/xGMI (gloo on CPU) gives the
average, then shared_half_s and the EMA with the previous global run locally
and identically on all ranks.

The reference averages the full state_dict, frozen tensors included
(utils/fed_utils.py:76-86); frozen tensors are identical on every client, so
their weighted mean is the tensor itself up to one rounding — they are left
untouched here.

On a GPU the three elementwise passes are HIP kernels (ffm_scale_by,
ffm_fedavg_finish); for BASELINE.json configs[0] (CPU/gloo plumbing, no GPU)
the same three passes run as host tensor arithmetic.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def is_group_s_block(key: str, shape: Sequence[int], num_groups: int) -> bool:
    """The per-group singular-value blocks that get the per-attribute weights and shared_half_s
    (utils/fed_utils.py:76,90: 'lora_S' in key and shape[0] == number of groups).  Only the real 2-D [G, r]
    tensors count: the 1-D lora_S_global [r] (GLOBAL_S) and SVLoRA's lora_S [r] also carry 'lora_S' in their
    keys and have shape[0] == G whenever rank == num_groups; taken for a block they would be read G*r wide,
    over the neighbouring tensors of the flat buffer."""
    return "lora_S" in key and len(shape) == 2 and shape[0] == num_groups


def element_weights(offsets: Dict[str, Tuple[int, Tuple[int, ...]]], numel: int, client: int,
                    participants: Sequence[int], n_client: Sequence[int],
                    n_client_by_attr: Optional[Sequence[Sequence[int]]]) -> Tensor:
    """Per-element FedAvg weight of `client` over the flat buffer (fp32, CPU)."""
    w = torch.zeros(numel, dtype=torch.float32)
    if client not in participants:
        return w                                            # frac < 1: non-selected ranks contribute 0
    total = float(sum(n_client[u] for u in participants))
    f = torch.tensor(n_client[client] / total, dtype=torch.float32)
    w.fill_(float(f))
    if n_client_by_attr is not None:
        by = torch.tensor(n_client_by_attr)
        tot = by[list(participants)].sum(0)
        fg = (by[client] / tot).to(torch.float32)           # same fp32 division as the reference
        G = by.shape[1]
        for key, (off, shp) in offsets.items():
            if is_group_s_block(key, shp, G):
                r = shp[1]
                w[off:off + G * r] = fg[:, None].expand(G, r).reshape(-1)
    return w


class FedAvgAggregator:
    """The round boundary of ONE flat buffer (the trainable tensors; RN50: a second instance over the BatchNorm buffers).

    A round is ``begin()``, one ``add(src, client, ...)`` per client this rank trained (none on a rank that trained none:
    it contributes zeros), ``finish(epoch, max_epoch)``: one all_reduce(SUM), shared_half_s, EMA with the previous global.
    ``aggregate()`` is the one-client-per-rank shorthand.  Both drivers of the product (``federated.run_fedotplora_ranks``)
    and ``bench.py --gpus N`` go through here, so the bench times the product's round boundary.  On a GPU every pass is a
    HIP kernel (ffm_scale_by / ffm_scale_acc / ffm_fedavg_finish) over device-resident weight vectors that are built once
    per (client, participants) and cached; on the CPU (BASELINE.json configs[0]: gloo plumbing) the same passes are host
    tensor arithmetic."""

    def __init__(self, flat: Tensor, offsets: Dict[str, Tuple[int, Tuple[int, ...]]], num_groups: int, rank_dim: int,
                 shared_half_s: bool = True, beta: float = 0.999, group=None, wcache_entries: int = 64):
        self.flat, self.offsets = flat, offsets
        self.G, self.r = num_groups, rank_dim
        self.shared_half_s, self.beta, self.group = shared_half_s, beta, group
        self.global_prev = flat.detach().clone()            # w_g: the global weights before round 0
        self.buf = torch.empty_like(flat)
        self._wcache: "Dict[tuple, Tensor]" = {}
        self._wcache_entries = wcache_entries
        self._added = 0
        self._grouped = False
        self.s_offsets = torch.tensor(
            [off for k, (off, shp) in offsets.items() if is_group_s_block(k, shp, num_groups)],
            dtype=torch.int64, device=flat.device)

    def _weights(self, client: int, participants: Sequence[int], n_client: Sequence[int],
                 n_client_by_attr: Optional[Sequence[Sequence[int]]]) -> Tensor:
        key = (client, tuple(participants), tuple(n_client),
               None if n_client_by_attr is None else tuple(map(tuple, n_client_by_attr)))
        w = self._wcache.get(key)
        if w is None:                                       # the counts do not change between rounds: built once
            if len(self._wcache) >= self._wcache_entries:
                self._wcache.pop(next(iter(self._wcache)))
            w = element_weights(self.offsets, self.flat.numel(), client, participants, n_client,
                                n_client_by_attr).to(self.flat.device)
            self._wcache[key] = w
        return w

    @torch.no_grad()
    def begin(self) -> None:
        self._added, self._grouped = 0, False

    @torch.no_grad()
    def add(self, src: Tensor, client: int, participants: Sequence[int], n_client: Sequence[int],
            n_client_by_attr: Optional[Sequence[Sequence[int]]] = None) -> None:
        """buf (+)= w_client (.) src: this rank's share of sum_k w_k theta_k (utils/fed_utils.py:76-86)."""
        w = self._weights(client, participants, n_client, n_client_by_attr)
        if src.is_cuda:
            from . import ops
            (ops.scale_by if self._added == 0 else ops.scale_acc)(src, w, self.buf)
        elif self._added == 0:
            torch.mul(src, w, out=self.buf)
        else:
            self.buf.add_(src * w)
        self._added += 1
        self._grouped = self._grouped or n_client_by_attr is not None

    @torch.no_grad()
    def finish(self, epoch: int, max_epoch: int, grouped: Optional[bool] = None) -> Tensor:
        """All ranks call this once per round; returns the new global weights (``global_prev``, updated in place;
        utils/fed_utils.py:88-98: w = (1-b)*avg + b*w_g with b = beta * epoch / max_epoch).  ``grouped``: whether the
        per-attribute counts were in play (the reference's guard on shared_half_s) - by default what ``add`` saw; a rank
        without a client of its own passes it explicitly."""
        if self._added == 0:
            self.buf.zero_()
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
        beta_decay = self.beta * (epoch / max(max_epoch, 1))
        use_half = self.shared_half_s and (self._grouped if grouped is None else grouped) and self.s_offsets.numel() > 0
        if self.buf.is_cuda:
            from . import ops
            ops.fedavg_finish(self.buf, self.global_prev, self.global_prev, self.s_offsets if use_half else None,
                              self.G, self.r, use_half, float(beta_decay))
        else:
            if use_half:
                G, r = self.G, self.r
                for off in self.s_offsets.tolist():
                    blk = self.buf[off:off + G * r].view(G, r)
                    blk[:, : r // 2] = blk[:, : r // 2].mean(0, keepdim=True)
            self.global_prev.copy_((1 - beta_decay) * self.buf + beta_decay * self.global_prev)
        self._added = 0
        return self.global_prev

    @torch.no_grad()
    def aggregate(self, client: int, participants: Sequence[int], n_client: Sequence[int],
                  n_client_by_attr: Optional[Sequence[Sequence[int]]], epoch: int, max_epoch: int) -> Tensor:
        """One client per rank: all ranks call this at the round boundary; afterwards `flat` holds the new global
        weights on every rank."""
        self.begin()
        self.add(self.flat, client, participants, n_client, n_client_by_attr)
        self.flat.copy_(self.finish(epoch, max_epoch, grouped=n_client_by_attr is not None))
        return self.flat
