"""BASELINE.json configs[0]: 2-client FedAvg plumbing on CPU over gloo.

Each rank holds one client's flat trainable buffer; the aggregate must equal the
oracle's restatement of average_weights_EMA (utils/fed_utils.py:42-100) applied
to the per-client state_dicts, and the committed golden vectors for the same
routine (tests/test_oracle_golden.py pins the oracle itself)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fairfedmed_amd import config as C
from fairfedmed_amd import synth
from fairfedmed_amd.fedavg import FedAvgAggregator, element_weights


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layout(mcfg):
    """Flat layout without touching the GPU (FlatParams allocates on a device)."""
    shapes = synth.manifest(mcfg)
    offsets, off = {}, 0
    for k in synth.trainable_keys(mcfg):
        shp = tuple(shapes[k])
        n = 1
        for s in shp:
            n *= s
        offsets[k] = (off, shp)
        off = (off + n + 3) // 4 * 4
    return offsets, off


def _client_state(mcfg, client, rnd):
    sd = synth.make_state_dict(mcfg, seed=100 + 10 * rnd + client, lora_init="random")
    return {k: sd[k] for k in synth.trainable_keys(mcfg)}


def _flatten(state, offsets, numel):
    flat = torch.zeros(numel)
    for k, (off, shp) in offsets.items():
        flat[off:off + state[k].numel()] = state[k].reshape(-1)
    return flat


N_CLIENT = [120, 80]
BY_ATTR = [[60, 40, 20], [10, 30, 40]]
ROUNDS = 3


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mcfg = C.vit_tiny(rank=4)
    offsets, numel = _layout(mcfg)
    init = _flatten(_client_state(mcfg, 99, 0), offsets, numel)      # common initial global weights
    flat = init.clone()
    agg = FedAvgAggregator(flat, offsets, mcfg.lora.num_groups, mcfg.lora.rank, shared_half_s=True)
    for rnd in range(ROUNDS):
        flat.copy_(_flatten(_client_state(mcfg, rank, rnd), offsets, numel))   # "local training" result
        agg.aggregate(rank, [0, 1], N_CLIENT, BY_ATTR, rnd, ROUNDS)
    torch.save(flat, os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_client_fedavg_over_gloo():
    from oracle import fairlora_oracle as O
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        got = [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(2)]
    assert torch.equal(got[0], got[1]), "ranks disagree on the new global weights"
    mcfg = C.vit_tiny(rank=4)
    offsets, numel = _layout(mcfg)
    w_g = _client_state(mcfg, 99, 0)
    for rnd in range(ROUNDS):
        w = {c: _client_state(mcfg, c, rnd) for c in range(2)}
        w_g = O.average_weights_ema(w_g, w, [0, 1], N_CLIENT, BY_ATTR, rnd, ROUNDS, shared_half_s=True)
    ref = _flatten(w_g, offsets, numel)
    assert torch.allclose(got[0], ref, rtol=1e-6, atol=1e-7)


def test_element_weights_rows_and_nonparticipant():
    mcfg = C.vit_tiny(rank=4)
    offsets, numel = _layout(mcfg)
    w = element_weights(offsets, numel, 0, [0, 1], N_CLIENT, BY_ATTR)
    k = next(k for k in offsets if k.endswith("lora_S.weight"))
    off, (G, r) = offsets[k]
    assert torch.allclose(w[off:off + G * r].view(G, r)[:, 0], torch.tensor([60 / 70, 40 / 70, 20 / 60]))
    assert abs(float(w[0]) - 0.6) < 1e-7
    assert float(element_weights(offsets, numel, 1, [0], N_CLIENT, BY_ATTR).abs().max()) == 0.0


def test_single_process_matches_oracle_without_attr_counts():
    """No dist group: a 1-client 'average' (local mode) with plain weights and no shared_half_s."""
    from oracle import fairlora_oracle as O
    mcfg = C.vit_tiny(rank=4)
    offsets, numel = _layout(mcfg)
    w_g = _client_state(mcfg, 99, 0)
    flat = _flatten(w_g, offsets, numel)
    agg = FedAvgAggregator(flat, offsets, 3, 4, shared_half_s=True)
    loc = _client_state(mcfg, 0, 0)
    flat.copy_(_flatten(loc, offsets, numel))
    agg.aggregate(0, [0], [10], None, 2, 4)
    ref = O.average_weights_ema(w_g, {0: loc}, [0], [10], None, 2, 4, shared_half_s=True)
    assert torch.allclose(flat, _flatten(ref, offsets, numel), rtol=1e-6, atol=1e-7)
