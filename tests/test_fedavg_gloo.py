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


# ----------------------------------------------------------------------------------------------------------------
# fairfedmed_amd.federated.run_fedotplora_ranks over gloo, world 2: three clients dealt to two ranks, "training" is a
# deterministic edit of the flat buffer.  The result must equal the one-process driver (list-of-clients average).
# ----------------------------------------------------------------------------------------------------------------
class _FlatTrainer:
    """Stand-in for the HIP trainer: engine.params.{flat, offsets}, train(idx) edits the flat buffer."""

    def __init__(self, users):
        from types import SimpleNamespace as NS
        mcfg = C.vit_tiny(rank=4)
        offsets, numel = _layout(mcfg)
        flat = _flatten(_client_state(mcfg, 99, 0), offsets, numel)
        self.engine = NS(params=NS(flat=flat, offsets=offsets, keys=list(offsets)), cfg=mcfg)
        self.mom, self.steps = torch.zeros_like(flat), 0
        self.cfg = NS(DATASET=NS(USERS=users, ATTRIBUTE_TYPE="race"))
        self.fed_train_loader_x_dict = {
            i: NS(dataset=type("D", (), {"__len__": lambda s, n=40 * (i + 1): n,
                                         "count_by_attribute": lambda s, a, i=i: [10 + i, 20, 5 * (i + 1)]})())
            for i in range(users)}
        outer = self

        class _Model:
            def state_dict(self):
                p = outer.engine.params
                return {k: p.flat[o:o + int(torch.tensor(s).prod())].view(s) for k, (o, s) in p.offsets.items()}

            def load_state_dict(self, sd, strict=True):
                p = outer.engine.params
                for k, v in sd.items():
                    o, s = p.offsets[k]
                    p.flat[o:o + v.numel()] = v.reshape(-1)
        self.model = _Model()

    def fed_before_train(self): pass
    def fed_after_train(self): pass

    def train(self, idx, global_epoch, is_fed, is_last_client):
        f = self.engine.params.flat
        f.mul_(1.0 + 0.01 * (idx + 1)).add_(0.001 * (global_epoch + 1) * (idx + 1))
        if self.shared_opt:
            # a momentum-like state that outlives the client, as the reference's single optimizer does (quirk 8)
            self.mom.mul_(0.9).add_(0.01 * (idx + 1))
            self.steps += 1
            f.add_(self.mom * (0.5 ** (self.steps // 4)))

    shared_opt = False

    def optimizer_state(self):
        return self.mom, torch.tensor([self.steps], dtype=torch.float64)

    def load_optimizer_state(self, mom, scal):
        if mom.data_ptr() != self.mom.data_ptr():
            self.mom.copy_(mom)
        self.steps = int(scal[0])

    def test(self, idx, current_epoch):
        return [float(self.engine.params.flat.mean()), 0.0, 0.0, 0.5]


def _fed_worker(rank, world, port, outdir):
    from fairfedmed_amd import federated as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = F.FedArgs(num_users=3, frac=0.7, round=3, shared_half_s=True, seed=5)
    hist = F.run_fedotplora_ranks(_FlatTrainer(3), args, log=lambda *_: None)
    torch.save({"flat": hist["global_flat"], "acc": hist["acc"]}, os.path.join(outdir, f"fed{rank}.pt"))
    dist.destroy_process_group()


def test_round_loop_over_two_ranks_equals_the_single_process_driver():
    from fairfedmed_amd import federated as F
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_fed_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        got = [torch.load(os.path.join(d, f"fed{r}.pt")) for r in range(2)]
    assert torch.equal(got[0]["flat"], got[1]["flat"])
    tr = _FlatTrainer(3)
    hist = F.run_fedotplora(tr, F.FedArgs(num_users=3, frac=0.7, round=3, shared_half_s=True, seed=5), log=lambda *_: None)
    p = tr.engine.params
    for k, v in hist["global_weights"].items():
        o, s = p.offsets[k]
        ref = v.reshape(-1)
        assert torch.allclose(got[0]["flat"][o:o + ref.numel()], ref, rtol=1e-5, atol=1e-7), k
    assert all(abs(a - b) < 1e-6 for a, b in zip(got[0]["acc"], hist["acc"]))


def _compat_worker(rank, world, port, outdir, compat):
    from fairfedmed_amd import federated as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = F.FedArgs(num_users=3, frac=0.7, round=3, shared_half_s=True, seed=5, compat_sequential_optimizer=compat)
    tr = _FlatTrainer(3)
    tr.shared_opt = True
    hist = F.run_fedotplora_ranks(tr, args, log=lambda *_: None)
    torch.save({"flat": hist["global_flat"]}, os.path.join(outdir, f"c{int(compat)}_{rank}.pt"))
    dist.destroy_process_group()


def test_compat_sequential_optimizer_reproduces_the_shared_optimizer_across_ranks():
    """with compat_sequential_optimizer the two-rank run hands the optimizer state from client
    to client in reference order and equals the one-process (one shared optimizer) run; without it the ranks keep
    their own optimizer state and the result differs."""
    from fairfedmed_amd import federated as F
    tr = _FlatTrainer(3)
    tr.shared_opt = True
    hist = F.run_fedotplora(tr, F.FedArgs(num_users=3, frac=0.7, round=3, shared_half_s=True, seed=5), log=lambda *_: None)
    p = tr.engine.params
    ref = torch.zeros_like(p.flat)
    for k, v in hist["global_weights"].items():
        o, s = p.offsets[k]
        ref[o:o + v.numel()] = v.reshape(-1)
    with tempfile.TemporaryDirectory() as d:
        for compat in (True, False):
            mp.spawn(_compat_worker, args=(2, _free_port(), d, compat), nprocs=2, join=True)
        on = [torch.load(os.path.join(d, f"c1_{r}.pt"))["flat"] for r in range(2)]
        off = [torch.load(os.path.join(d, f"c0_{r}.pt"))["flat"] for r in range(2)]
    assert torch.equal(on[0], on[1]) and torch.equal(off[0], off[1])
    used = ref != 0
    assert torch.allclose(on[0][used], ref[used], rtol=1e-5, atol=1e-7)
    assert not torch.allclose(off[0][used], ref[used], rtol=1e-3, atol=1e-5)


def _pers_worker(rank, world, port, outdir):
    from fairfedmed_amd import federated as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = F.FedArgs(num_users=3, round=3, shared_half_s=True, seed=5, idxs_users_train=[0, 2], local_s=True,
                     avg_prompt=1, num_prompt=2)
    hist = F.run_fedotplora_ranks(_FlatTrainer(3), args, log=lambda *_: None)
    torch.save({"flat": hist["global_flat"], "acc": hist["acc"]}, os.path.join(outdir, f"p{rank}.pt"))
    dist.destroy_process_group()


def test_personalised_clients_over_two_ranks_equal_the_single_process_driver():
    """--idxs_users_train with LOCAL_S (federated_main.py:645-652): only clients 0 and 2 train, each keeps its own local
    prompts ctx[avg_prompt:] and lora_S; every client is evaluated with its personalised weights.  The rank-parallel
    driver must report the same per-round scores (they depend on the personalised weights) and the same global."""
    from fairfedmed_amd import federated as F
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_pers_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        got = [torch.load(os.path.join(d, f"p{r}.pt")) for r in range(2)]
    assert torch.equal(got[0]["flat"], got[1]["flat"]) and got[0]["acc"] == got[1]["acc"]
    tr = _FlatTrainer(3)
    args = F.FedArgs(num_users=3, round=3, shared_half_s=True, seed=5, idxs_users_train=[0, 2], local_s=True,
                     avg_prompt=1, num_prompt=2)
    hist = F.run_fedotplora(tr, args, log=lambda *_: None)
    p = tr.engine.params
    for k, v in hist["global_weights"].items():
        o, s = p.offsets[k]
        ref = v.reshape(-1)
        assert torch.allclose(got[0]["flat"][o:o + ref.numel()], ref, rtol=1e-5, atol=1e-7), k
    assert all(abs(a - b) < 1e-6 for a, b in zip(got[0]["acc"], hist["acc"])), (got[0]["acc"], hist["acc"])


# ----------------------------------------------------------------------------------------------------------------
# (1) with an unseeded numpy generator every rank used to draw ITS OWN client subset; (2) each rank's StepLR
# counted only its own client-epochs.  Rank 0's draw is now broadcast, and every rank positions the scheduler from the
# global count of client-epochs, so both are independent of the world size.
# ----------------------------------------------------------------------------------------------------------------
class _SchedTrainer(_FlatTrainer):
    """_FlatTrainer with the trainer's scheduler surface: one shared StepLR advanced `steps_per_update()` times per local
    epoch (the reference's two registered names), train() logs where the schedule stood when each client started."""
    max_epoch = 1

    def __init__(self, users):
        super().__init__(users)
        from types import SimpleNamespace as NS
        self.sched = NS(last_epoch=0, step_size=3, gamma=0.1)
        self.log = []

    def steps_per_update(self):
        return 2

    def set_lr_epoch(self, n):
        self.sched.last_epoch = int(n)

    def train(self, idx, global_epoch, is_fed, is_last_client):
        self.log.append((global_epoch, idx, self.sched.last_epoch))
        lr = self.sched.gamma ** (self.sched.last_epoch // self.sched.step_size)
        self.engine.params.flat.mul_(1.0 + 0.01 * lr * (idx + 1))
        self.sched.last_epoch += self.max_epoch * self.steps_per_update()


def _sched_worker(rank, world, port, outdir, seeded):
    import numpy as np
    from fairfedmed_amd import federated as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np.random.seed(1000 + rank)                       # the ranks' generators disagree unless args.seed aligns them
    args = F.FedArgs(num_users=4, frac=0.5, round=4, shared_half_s=True, seed=5 if seeded else None)
    tr = _SchedTrainer(4)
    hist = F.run_fedotplora_ranks(tr, args, log=lambda *_: None)
    torch.save({"flat": hist["global_flat"], "log": tr.log, "last_epoch": tr.sched.last_epoch,
                "keys": {i: sorted(w) for i, w in hist["local_weights_per"].items()}},
               os.path.join(outdir, f"s{int(seeded)}_{rank}.pt"))
    dist.destroy_process_group()


def test_ranks_share_one_client_draw_and_one_lr_schedule():
    from fairfedmed_amd import federated as F
    with tempfile.TemporaryDirectory() as d:
        for seeded in (False, True):
            mp.spawn(_sched_worker, args=(2, _free_port(), d, seeded), nprocs=2, join=True)
        un = [torch.load(os.path.join(d, f"s0_{r}.pt")) for r in range(2)]
        se = [torch.load(os.path.join(d, f"s1_{r}.pt")) for r in range(2)]
    # unseeded: both ranks still aggregate the SAME participants (rank 0's draw) and end with identical weights
    assert torch.equal(un[0]["flat"], un[1]["flat"]) and un[0]["last_epoch"] == un[1]["last_epoch"]
    for rnd in range(1, 4):
        trained = sorted(i for r in un for (e, i, _) in r["log"] if e == rnd)
        assert len(trained) == 2 and len(set(trained)) == 2, (rnd, trained)       # frac 0.5 of 4: two DISTINCT clients
    # seeded: the schedule position every client started from equals the one-process driver's (one shared scheduler
    # advanced by every client in turn), and so do the final weights
    tr = _SchedTrainer(4)
    hist = F.run_fedotplora(tr, F.FedArgs(num_users=4, frac=0.5, round=4, shared_half_s=True, seed=5), log=lambda *_: None)
    both = sorted(se[0]["log"] + se[1]["log"])
    assert both == sorted(tr.log), (both, tr.log)
    assert se[0]["last_epoch"] == se[1]["last_epoch"] == tr.sched.last_epoch
    p = tr.engine.params
    for k, v in hist["global_weights"].items():
        o, s = p.offsets[k]
        assert torch.allclose(se[0]["flat"][o:o + v.numel()], v.reshape(-1), rtol=1e-5, atol=1e-7), k
    # the per-client weights the CLI saves exist on every rank, for every client, under the trainable keys
    assert sorted(se[0]["keys"]) == [0, 1, 2, 3] and se[0]["keys"][0] == sorted(p.offsets)


# ----------------------------------------------------------------------------------------------------------------
# GLOBAL_S with rank == num_groups: the 1-D lora_S_global [r] has shape[0] == G.  Neither driver may treat it as a
# [G, r] block (the flat-buffer view would average over the neighbouring lora_B entries).
# ----------------------------------------------------------------------------------------------------------------
class _GlobalSTrainer(_FlatTrainer):
    def __init__(self, users):
        import dataclasses
        from types import SimpleNamespace as NS
        super().__init__(users)
        base = C.vit_tiny(rank=2, num_groups=2)                        # 2 gender groups, rank 2
        mcfg = dataclasses.replace(base, lora=dataclasses.replace(base.lora, global_s=True))
        offsets, numel = _layout(mcfg)
        assert any(k.endswith("lora_S_global.weight") and s == (2,) for k, (_, s) in offsets.items())
        self.fed_train_loader_x_dict = {
            i: NS(dataset=type("D", (), {"__len__": lambda s, n=40 * (i + 1): n,
                                         "count_by_attribute": lambda s, a, i=i: [10 + i, 5 * (i + 1)]})())
            for i in range(users)}
        flat = _flatten(_client_state(mcfg, 99, 0), offsets, numel)
        self.engine = NS(params=NS(flat=flat, offsets=offsets, keys=list(offsets)), cfg=mcfg)
        self.mom = torch.zeros_like(flat)


def _gs_worker(rank, world, port, outdir):
    from fairfedmed_amd import federated as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = F.FedArgs(num_users=3, frac=1.0, round=2, shared_half_s=True, seed=5)
    hist = F.run_fedotplora_ranks(_GlobalSTrainer(3), args, log=lambda *_: None)
    torch.save({"flat": hist["global_flat"]}, os.path.join(outdir, f"gs{rank}.pt"))
    dist.destroy_process_group()


def test_global_s_with_rank_equal_to_num_groups_over_two_ranks():
    from fairfedmed_amd import federated as F
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_gs_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        got = [torch.load(os.path.join(d, f"gs{r}.pt")) for r in range(2)]
    assert torch.equal(got[0]["flat"], got[1]["flat"])
    tr = _GlobalSTrainer(3)
    hist = F.run_fedotplora(tr, F.FedArgs(num_users=3, frac=1.0, round=2, shared_half_s=True, seed=5), log=lambda *_: None)
    p = tr.engine.params
    for k, v in hist["global_weights"].items():
        o, s = p.offsets[k]
        assert tuple(v.shape) == tuple(s), k                           # no [r] -> [G, r] broadcast
        ref = v.reshape(-1)
        assert torch.allclose(got[0]["flat"][o:o + ref.numel()], ref, rtol=1e-5, atol=1e-7), k
